#!/usr/bin/env python3
"""bench.py — Lizard block-compress throughput on MI355X (BASELINE.json metric).

A "step" is one pass of the hot path over one batch: every rank compresses its own `--blocks`
independent blocks of `--block-size` bytes of datagen-P50 synthetic input that is already resident in
HBM (block b of rank r is RDG_genBuffer(blockSize, 0.5, seed = r*blocks + b)), then (N > 1) the ranks
exchange the per-block compressed sizes with one RCCL all-gather so that every rank can compute global
output offsets.  Default workload = BASELINE.json configs[1]: level 10, 65 536 x 256 KiB per GPU.

  python bench.py --gpus 1 --steps 3 --warmup 1
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line. `value` is whole-job input MB/s (MB = 10^6 B, as reference
programs/bench.c:253-255) over the barrier-bracketed timed region, max over ranks.
`roofline.achieved` = algorithmic bytes per launch (input read once + compressed output written once,
SURVEY.md §8d) / average kernel duration measured with HIP events on the launch stream.
`cpu_baseline` = the reference CPU compressor (oracle/_ref, kind "reference") or its restatement
(oracle/, kind "port") timed on this box's host, one thread, on a bounded sample of the same blocks; the
same leg checks a sample of the launch's output blocks bit for bit against the oracle
(`cpu_baseline.gpu_blocks_checked_bit_exact`).  Nothing else in this file touches oracle/.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md


KERNEL_OF_LEVEL = {10: "lz_fast12_kernel<false, true>", 30: "lz_fast12_kernel<true, true>", 11: "lz_fast18_kernel<false>",
                   31: "lz_fast18_kernel<true>", 21: "lz_pricefast14_kernel<false>", 41: "lz_pricefast14_kernel<true>",
                   22: "lz_pricefast18_kernel<false>", 42: "lz_pricefast18_kernel<true>"}
for _l in range(13, 18):       # hashChain rows: searchLength 5 for 13-15 / 34-36, 4 for 16-17 / 37-38
    KERNEL_OF_LEVEL[_l] = "lz_hashchain_kernel<false, %d>" % (5 if _l <= 15 else 4)
    KERNEL_OF_LEVEL[_l + 21] = "lz_hashchain_kernel<true, %d>" % (5 if _l <= 15 else 4)


def cpu_baseline(level, block_size, n_blocks, budget_s):
    """Time the CPU compressor on the first n_blocks blocks of rank 0's workload (host-generated with the
    same generator/seeds). Best-of-N full passes like the reference's bench (programs/bench.c:231-246)."""
    import util
    from lizard_amd import _lib
    L = _lib.lib()
    ref = None
    path = os.path.join(ROOT, "oracle", "_ref", "liblizard_ref.so")   # stock build: what `lizard -b` times
    if os.path.exists(path):
        ref = ctypes.CDLL(path)
        fn, kind = ref.Lizard_compress, "reference"
    else:
        fn, kind = util.oracle().lzo_compress, "port"
    fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    fn.restype = ctypes.c_int
    buf = ctypes.create_string_buffer(n_blocks * block_size)
    base = ctypes.addressof(buf)
    for b in range(n_blocks):
        L.LizardGPU_datagen_host(base + b * block_size, block_size, 0.5, 0.0, b)
    bound = util.oracle().lzo_compress_bound(block_size)
    out = ctypes.create_string_buffer(bound)
    best, total_c, t_start, loops = None, 0, time.perf_counter(), 0
    while True:
        t0 = time.perf_counter()
        total_c = 0
        for b in range(n_blocks):
            total_c += fn(base + b * block_size, out, block_size, bound, level)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
        loops += 1
        if time.perf_counter() - t_start > budget_s:
            break
    nbytes = n_blocks * block_size
    return {"value": round(nbytes / best / 1e6, 1), "unit": "MB/s", "cores": 1, "kind": kind,
            "sample": f"first {n_blocks} blocks x {block_size} B of rank 0's workload, level {level}, "
                      f"best of {loops} passes, 1 thread",
            "ratio": round(nbytes / total_c, 4), "host_cpus": os.cpu_count()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--level", type=int, default=10)
    ap.add_argument("--block-size", type=int, default=262144)
    ap.add_argument("--blocks", type=int, default=65536, help="blocks per GPU (weak scaling)")
    ap.add_argument("--verify", type=int, default=48, help="GPU output blocks the cpu_baseline leg checks bit-exact against the oracle")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--cpu-blocks", type=int, default=256)
    ap.add_argument("--no-cpu", action="store_true")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist
    from lizard_amd import _lib, api

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)       # "nccl" is RCCL on ROCm

    L = _lib.lib()
    L.LizardGPU_setDevice(local_rank)
    if not L.LizardGPU_levelSupported(args.level):
        raise SystemExit(f"level {args.level} is not implemented on the GPU path")
    nb, bs = args.blocks, args.block_size
    stride = (api.Lizard_compressBound(bs) + 63) & ~63
    src = torch.empty(nb * bs, dtype=torch.uint8, device=dev)
    dst = torch.empty(nb * stride, dtype=torch.uint8, device=dev)
    sizes = torch.zeros(nb, dtype=torch.int32, device=dev)
    from lizard_amd.sharding import gather_block_sizes
    gathered = None
    stream = torch.cuda.current_stream(dev)
    _lib.check(L.LizardGPU_datagen_device(src.data_ptr(), nb, bs, 0.5, 0.0, rank * nb, ctypes.c_void_p(stream.cuda_stream)),
               "LizardGPU_datagen_device")
    torch.cuda.synchronize()

    kernel_ms = []

    def step(timed):
        nonlocal gathered
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        api.compress_blocks_device(src, bs, args.level, dst=dst, sizes=sizes)
        e1.record(stream)
        if world > 1:                                         # RCCL over xGMI: 4 B per block per rank
            gathered = gather_block_sizes(sizes, world * nb)  # -> sizes + global output offsets on every rank
        if timed:
            kernel_ms.append((e0, e1))

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step(False)
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    kms = [a.elapsed_time(b) for a, b in kernel_ms]
    in_bytes = nb * bs
    out_bytes = int(sizes.to(torch.int64).sum().item())
    tot_in, tot_out = in_bytes * world, out_bytes
    if world > 1:
        all_sizes, offsets = gathered
        tot_out = int(all_sizes.to(torch.int64).sum().item())
        assert int(offsets[-1].item()) + int(all_sizes[-1].item()) == tot_out

    def check_sample():
        """Checker of the cpu_baseline leg: a sample of this launch's blocks against the CPU oracle, bit for bit."""
        import util
        n_ok = 0
        idx = sorted(set(list(range(min(nb, args.verify // 2))) + [int(i) for i in np.linspace(0, nb - 1, args.verify // 2)]))
        sz = sizes.cpu().numpy()
        for b in idx:
            got = dst[b * stride:b * stride + int(sz[b])].cpu().numpy().tobytes()
            blk = ctypes.create_string_buffer(bs)
            L.LizardGPU_datagen_host(blk, bs, 0.5, 0.0, b)
            assert bytes(src[b * bs:(b + 1) * bs].cpu().numpy()) == blk.raw, f"device datagen differs from host datagen at block {b}"
            want = util.oracle_compress(blk.raw, args.level)
            assert got == want, f"block {b}: GPU output differs from the oracle"
            n_ok += 1
        return n_ok

    if rank == 0:
        avg_k = sum(kms) / len(kms) / 1e3
        alg_bytes = in_bytes + out_bytes                     # per launch on this GPU
        traffic = None                                       # HBM bytes per launch from the committed PMC passes, if this config was profiled
        try:
            with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
                for ent in json.load(f)["entries"]:
                    if (ent["level"], ent["block_size"], ent["blocks_per_gpu"]) == (args.level, bs, nb):
                        traffic = ent["traffic_bytes"]        # the LAST matching entry (latest round) wins
        except (OSError, KeyError, ValueError):
            pass
        res = {
            "metric": "compress MB/s (input), 256 KiB blocks level -10" if (args.level, bs) == (10, 262144)
                      else f"compress MB/s (input), {bs} B blocks level -{args.level}",
            "value": round(tot_in * args.steps / elapsed / 1e6, 1),
            "unit": "MB/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": f"level -{args.level}, {nb} x {bs} B independent blocks per GPU, datagen P50 "
                                   f"(block b = RDG_genBuffer(seed b)), inputs resident in HBM",
                       "level": args.level, "block_size": bs, "blocks_per_gpu": nb,
                       "resident_waves": int(L.LizardGPU_residentWaves())},
            "ratio": round(tot_in / tot_out, 4),
            "compressed_bytes": tot_out,
            "roofline": {"bound": "hbm", "achieved": round(alg_bytes / avg_k / 1e9, 2), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(alg_bytes / avg_k / 1e9 / HBM_PEAK_GBS, 5), "traffic": traffic,
                         "kernel": KERNEL_OF_LEVEL.get(args.level, "?"), "avg_kernel_ms": round(avg_k * 1e3, 3),
                         "algorithmic_bytes_per_launch": alg_bytes},
        }
        if world == 1 and not args.no_cpu:      # the CPU leg: the only place the oracle / oracle/_ref is touched
            res["cpu_baseline"] = cpu_baseline(args.level, bs, min(args.cpu_blocks, nb), args.cpu_seconds)
            res["cpu_baseline"]["gpu_blocks_checked_bit_exact"] = check_sample() if args.verify > 0 else 0
            res["speedup_vs_cpu_1core"] = round(res["value"] / res["cpu_baseline"]["value"], 2)
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
