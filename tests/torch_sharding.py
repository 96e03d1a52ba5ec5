"""TEST INFRASTRUCTURE: the torch.distributed twin of the library's size gather (lizard_amd/csrc/lizard_shard_core.h is the product's
multi-GPU path; this form exists for the gloo world-size-2 CPU test and as the cross-check of the C partition code).

Multi-GPU sharding of independent Lizard blocks.

Blocks share nothing, so ranks own contiguous block ranges and the only exchange on the path is ONE
all-gather of the per-block compressed sizes (uint32 per block), after which every rank computes the
global exclusive prefix sum = byte offset of every block in the concatenated output.  One process per
GPU; backend "nccl" (= RCCL over xGMI) on MI355X, "gloo" in the CPU tests.
"""
import torch
import torch.distributed as dist


def shard_range(n_blocks, rank, world):
    """Contiguous, balanced partition: rank r owns [start, start+count)."""
    base, rem = divmod(n_blocks, world)
    start = rank * base + min(rank, rem)
    return start, base + (1 if rank < rem else 0)


def gather_block_sizes(local_sizes, n_blocks, group=None):
    """All-gather per-block compressed sizes (int32 tensor of this rank's shard, in block order).
    Returns (sizes[n_blocks] int32, offsets[n_blocks] int64) identical on every rank."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        sizes = local_sizes
    else:
        counts = [shard_range(n_blocks, r, world)[1] for r in range(world)]
        width = max(counts)
        pad = torch.zeros(width, dtype=local_sizes.dtype, device=local_sizes.device)
        pad[:local_sizes.numel()] = local_sizes
        out = torch.empty(world * width, dtype=local_sizes.dtype, device=local_sizes.device)
        dist.all_gather_into_tensor(out, pad, group=group)
        sizes = torch.cat([out[r * width:r * width + counts[r]] for r in range(world)])
    s64 = sizes.to(torch.int64)
    return sizes, torch.cumsum(s64, 0) - s64
