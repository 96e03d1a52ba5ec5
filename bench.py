#!/usr/bin/env python3
"""bench.py — Lizard block-compress throughput on MI355X (BASELINE.json metric).

A "step" is one pass of the hot path over one batch: every rank compresses its own blocks of datagen-P50
synthetic input that is already resident in HBM (block b of rank r is RDG_genBuffer(blockSize, 0.5,
seed = r*blocks + b)), then (N > 1) the ranks exchange the per-block compressed sizes with one RCCL all-gather
(inside the library, LizardGPU_gatherSizes_device) so that every rank can compute global output offsets.

  python bench.py --gpus 1 --steps 3 --warmup 1
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  Its top-level fields are BASELINE.json configs[1] (level 10, 65 536 x 256 KiB per
GPU): `value` = whole-job input MB/s (MB = 10^6 B, as reference programs/bench.c:253-255) over the
barrier-bracketed timed region of exactly --steps steps, max over ranks.  The same process then times the other
BASELINE configs with the same --steps / --warmup and reports them under "configs": level 21 and level 30 at
16 384 x 256 KiB, level 10 at 4 096 x 4 MiB (configs[2..4]); "config1" is configs[0], the reference's own
CPU-runnable case (64 MiB RDG_genBuffer P50 seed 0 in 256 KiB blocks, programs/bench.c's loop over Lizard_compress).
Per config:
  roofline      achieved = algorithmic bytes per launch (input read once + compressed output written once,
                SURVEY.md §8d) / mean kernel duration (HIP events on the launch stream); traffic = fabric-side bytes
                of one launch from the committed rocprofv3 PMC passes (profiles/pmc_traffic.json)
  cpu_baseline  the stock reference library (oracle/_ref/liblizard_ref.so, kind "reference"; the oracle restatement,
                kind "port", when it is absent) on one host core over a bounded sample of the same blocks
  blocks_checked / blocks_checked_bytes   after the timed region EVERY block's compressed size is compared with the
                zero-state reference (oracle/_ref/liblizard_ref_reset.so, else the oracle restatement) run on the host
                cores, and >= 2 048 blocks byte for byte
"end_to_end" is the PCIe-inclusive rate of the host-buffer entry (LizardGPU_compressBlocks_host_packed) on a 4 GiB sample
of the headline workload, from pageable and from pinned memory — never `value`.
Only the cpu_baseline / verification legs touch oracle/ (as the checker); the timed region calls the product library.

  python bench.py --level 21 --blocks 16384         one configuration only (profiling runs)
"""
import argparse
import concurrent.futures
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md

_SIG = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]


def kernel_name(level, bs):
    huf = "true" if level >= 30 else "false"
    base = level - 20 if level >= 30 else level
    if level >= 34 and level <= 38:
        base = level - 21
    if base == 10:
        return "lz_fast12_kernel<%s, %s>" % (huf, "true" if bs <= (4 << 20) else "false")
    if base == 11:
        return "lz_fast18_kernel<%s>" % huf
    if base == 21:
        return "lz_pricefast14_kernel<%s, %s>" % (huf, "true" if bs <= (256 << 10) else "false")
    if base == 22:
        return "lz_pricefast18_kernel<%s>" % huf
    return "lz_hashchain_kernel<%s, %d>" % (huf, 5 if base <= 15 else 4)


def load_checker(zero_state):
    """(fn, kind): the compiled reference when oracle/_ref travelled, else the oracle restatement."""
    import util
    name = "liblizard_ref_reset.so" if zero_state else "liblizard_ref.so"      # stock build: what `lizard -b` times
    path = os.path.join(ROOT, "oracle", "_ref", name)
    if os.path.exists(path):
        fn, kind = ctypes.CDLL(path).Lizard_compress, "reference"
    else:
        fn, kind = util.oracle().lzo_compress, "port"
    fn.argtypes = _SIG
    fn.restype = ctypes.c_int
    return fn, kind


def cpu_baseline(L, level, block_size, n_blocks, budget_s, seed0=0, whole_buffer=False):
    """Time the CPU compressor on n_blocks blocks (host-generated with the same generator/seeds; whole_buffer: ONE
    RDG_genBuffer of n_blocks*block_size bytes cut into blocks, BASELINE configs[0]).  Best-of-N full passes like the
    reference's bench (programs/bench.c:231-246), one thread."""
    import util
    fn, kind = load_checker(zero_state=False)
    buf = ctypes.create_string_buffer(n_blocks * block_size)
    base = ctypes.addressof(buf)
    if whole_buffer:
        L.LizardGPU_datagen_host(base, n_blocks * block_size, 0.5, 0.0, seed0)
    else:
        for b in range(n_blocks):
            L.LizardGPU_datagen_host(base + b * block_size, block_size, 0.5, 0.0, seed0 + b)
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
    what = (f"one RDG_genBuffer({nbytes} B, P50, seed {seed0}) in {n_blocks} blocks of {block_size} B" if whole_buffer
            else f"first {n_blocks} blocks x {block_size} B of rank 0's workload")
    return {"value": round(nbytes / best / 1e6, 1), "unit": "MB/s", "cores": 1, "kind": kind,
            "sample": f"{what}, level {level}, best of {loops} passes, 1 thread",
            "ratio": round(nbytes / total_c, 4), "compressed_bytes": total_c, "host_cpus": os.cpu_count()}, buf


def verify_all_blocks(L, level, bs, nb, src, dst, sizes, stride, seed0, byte_blocks, threads):
    """Checker leg: every block's size (and `byte_blocks` blocks' bytes) of the last launch against the zero-state
    reference run on the host cores.  Also pins the device generator to the host generator on a sample."""
    import numpy as np
    fn, kind = load_checker(zero_state=True)
    import util
    bound = util.oracle().lzo_compress_bound(bs)
    sz = sizes.cpu().numpy().astype(np.int64)
    step = max(1, nb // byte_blocks)
    chunk = max(1, min(nb, (512 << 20) // bs))
    tls = {}
    n_bytes_checked = 0

    def one(args):
        ptr, b = args
        import threading
        t = threading.get_ident()
        if t not in tls:
            tls[t] = ctypes.create_string_buffer(bound)
        out = tls[t]
        r = fn(ptr, out, bs, bound, level)
        want_bytes = (b % step == 0) or b < 64
        return b, r, (out.raw[:r] if want_bytes else None)

    with concurrent.futures.ThreadPoolExecutor(max_workers=threads) as pool:
        for c0 in range(0, nb, chunk):
            c1 = min(nb, c0 + chunk)
            host = src[c0 * bs:c1 * bs].cpu().numpy()
            base = host.ctypes.data
            for b, r, raw in pool.map(one, [(base + (b - c0) * bs, b) for b in range(c0, c1)]):
                assert r == int(sz[b]), f"level {level}: block {b}: GPU size {int(sz[b])} != reference size {r}"
                if raw is not None:
                    got = dst[b * stride:b * stride + r].cpu().numpy().tobytes()
                    assert got == raw, f"level {level}: block {b}: GPU bytes differ from the reference"
                    n_bytes_checked += 1
            if c0 == 0:                                     # device datagen == host datagen (what the workload claims to be)
                blk = ctypes.create_string_buffer(bs)
                for b in (0, 1, c1 - 1):
                    L.LizardGPU_datagen_host(blk, bs, 0.5, 0.0, seed0 + b)
                    assert host[(b - c0) * bs:(b - c0 + 1) * bs].tobytes() == blk.raw, f"device datagen differs from host datagen at block {b}"
    return nb, n_bytes_checked, kind


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--level", type=int, default=None, help="time ONE configuration only (with --block-size / --blocks)")
    ap.add_argument("--block-size", type=int, default=262144)
    ap.add_argument("--blocks", type=int, default=None, help="blocks per GPU (weak scaling)")
    ap.add_argument("--verify", type=int, default=2048, help="blocks per config checked byte for byte (sizes: all blocks); 0 = no check")
    ap.add_argument("--cpu-seconds", type=float, default=5.0, help="CPU baseline budget per config")
    ap.add_argument("--cpu-blocks", type=int, default=256)
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline / verification / end-to-end legs")
    ap.add_argument("--headline-only", action="store_true")
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
    _lib.check(L.LizardGPU_setDevice(local_rank), "LizardGPU_setDevice")

    # the RCCL size gather lives in the library (rccl.h); the launcher's transport only carries the 128-byte id
    gather_via = "none (1 GPU)"
    if world > 1:
        try:
            uid = torch.zeros(128, dtype=torch.uint8, device=dev)
            if rank == 0:
                buf = ctypes.create_string_buffer(128)
                _lib.check(L.LizardGPU_commUniqueId(buf), "LizardGPU_commUniqueId")
                uid.copy_(torch.frombuffer(bytearray(buf.raw), dtype=torch.uint8))
            dist.broadcast(uid, 0)
            _lib.check(L.LizardGPU_commInitRank(bytes(uid.cpu().numpy().tobytes()), world, rank), "LizardGPU_commInitRank")
            gather_via = "library: ncclAllGather via LizardGPU_gatherSizes_device"
        except Exception as e:                                  # keep the job alive on the collective torch already has
            gather_via = f"torch.distributed all_gather (library communicator unavailable: {e})"
            print(f"bench.py rank {rank}: {gather_via}", file=sys.stderr)

    if args.level is not None:
        plan = [(args.level, args.block_size, args.blocks or 16384)]
    else:
        plan = [(10, 262144, args.blocks or 65536)]
        if not args.headline_only:
            plan += [(21, 262144, 16384), (30, 262144, 16384), (10, 4 << 20, 6656)]
    max_in = max(nb * bs for _, bs, nb in plan)
    max_out = max(nb * ((api.Lizard_compressBound(bs) + 63) & ~63) for _, bs, nb in plan)
    src_all = torch.empty(max_in, dtype=torch.uint8, device=dev)
    dst_all = torch.empty(max_out, dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream(dev)
    from lizard_amd.sharding import gather_block_sizes
    threads = max(1, min(96, (os.cpu_count() or 8) - 2))
    state = {"gather": gather_via}

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def run_config(level, bs, nb, with_cpu):
        if not L.LizardGPU_levelSupported(level):
            raise SystemExit(f"level {level} is not implemented on the GPU path")
        stride = (api.Lizard_compressBound(bs) + 63) & ~63
        src, dst = src_all[:nb * bs], dst_all[:nb * stride]
        sizes = torch.zeros(nb, dtype=torch.int32, device=dev)
        all_sizes = torch.zeros(world * nb, dtype=torch.int32, device=dev)
        offsets = torch.zeros(world * nb + 1, dtype=torch.int64, device=dev)
        seed0 = rank * nb
        _lib.check(L.LizardGPU_datagen_device(src.data_ptr(), nb, bs, 0.5, 0.0, seed0, ctypes.c_void_p(stream.cuda_stream)),
                   "LizardGPU_datagen_device")
        torch.cuda.synchronize()
        kernel_ms = []
        gathered = [None]

        def step(timed):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            api.compress_blocks_device(src, bs, level, dst=dst, sizes=sizes)
            e1.record(stream)
            if world > 1:                                     # RCCL over xGMI: 4 B per block per rank
                rc = -1
                if state["gather"].startswith("library"):
                    rc = L.LizardGPU_gatherSizes_device(sizes.data_ptr(), world * nb, all_sizes.data_ptr(), offsets.data_ptr(),
                                                        ctypes.c_void_p(stream.cuda_stream))
                    if rc == 0:
                        gathered[0] = (all_sizes, offsets[:-1])
                    else:                                     # keep the job alive on the collective torch already has (same on every rank)
                        state["gather"] = "torch.distributed all_gather (library gather failed: %s)" % L.LizardGPU_lastError().decode(errors="replace")
                        print(f"bench.py rank {rank}: {state['gather']}", file=sys.stderr)
                if rc != 0:
                    gathered[0] = gather_block_sizes(sizes, world * nb)
            if timed:
                kernel_ms.append((e0, e1))

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
            asz, offs = gathered[0]
            tot_out = int(asz.to(torch.int64).sum().item())
            assert int(offs[world * nb - 1].item()) + int(asz[-1].item()) == tot_out
            assert torch.equal(asz[rank * nb:(rank + 1) * nb], sizes)
        res = None
        if rank == 0:
            avg_k = sum(kms) / len(kms) / 1e3
            alg_bytes = in_bytes + out_bytes                     # per launch on this GPU
            traffic = None                                       # fabric bytes per launch from the committed PMC passes
            try:
                with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
                    for ent in json.load(f)["entries"]:
                        if (ent["level"], ent["block_size"], ent["blocks_per_gpu"]) == (level, bs, nb):
                            traffic = ent["traffic_bytes"]        # the LAST matching entry (latest round) wins
            except (OSError, KeyError, ValueError):
                pass
            res = {
                "level": level, "block_size": bs, "blocks_per_gpu": nb,
                "workload": f"level -{level}, {nb} x {bs} B independent blocks per GPU, datagen P50 "
                            f"(block b = RDG_genBuffer(seed b)), inputs resident in HBM",
                "value": round(tot_in * args.steps / elapsed / 1e6, 1), "unit": "MB/s",
                "ms_per_step": round(elapsed / args.steps * 1e3, 3),
                "ratio": round(tot_in / tot_out, 4), "compressed_bytes": tot_out,
                "roofline": {"bound": "hbm", "achieved": round(alg_bytes / avg_k / 1e9, 2), "peak": HBM_PEAK_GBS,
                             "unit": "GB/s", "frac": round(alg_bytes / avg_k / 1e9 / HBM_PEAK_GBS, 5), "traffic": traffic,
                             "traffic_over_algorithmic": round(traffic / alg_bytes, 2) if traffic else None,
                             "kernel": kernel_name(level, bs), "avg_kernel_ms": round(avg_k * 1e3, 3),
                             "algorithmic_bytes_per_launch": alg_bytes},
            }
            if with_cpu:                            # the CPU legs: the only place the oracle / oracle/_ref is touched
                ncpu = min(args.cpu_blocks if bs <= (1 << 20) else 32, nb)
                res["cpu_baseline"], _ = cpu_baseline(L, level, bs, ncpu, args.cpu_seconds)
                res["speedup_vs_cpu_1core"] = round(res["value"] / res["cpu_baseline"]["value"], 2)
                if args.verify > 0:
                    t0 = time.perf_counter()
                    n_sz, n_by, kind = verify_all_blocks(L, level, bs, nb, src, dst, sizes, stride, seed0, args.verify, threads)
                    res["blocks_checked"] = n_sz
                    res["blocks_checked_bytes"] = n_by
                    res["checker"] = f"{kind} (zero-state), {threads} host threads, {round(time.perf_counter() - t0, 1)} s"
        return res, (src, dst, sizes, stride)

    with_cpu = world == 1 and not args.no_cpu
    results = []
    for level, bs, nb in plan:
        r, _ = run_config(level, bs, nb, with_cpu)
        results.append(r)

    if rank == 0:
        head = results[0]
        is_headline = (head["level"], head["block_size"]) == (10, 262144)
        out = {
            "metric": "compress MB/s (input), 256 KiB blocks level -10" if is_headline
                      else f"compress MB/s (input), {head['block_size']} B blocks level -{head['level']}",
            "value": head["value"], "unit": "MB/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": head["ms_per_step"],
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": head["workload"], "level": head["level"], "block_size": head["block_size"],
                       "blocks_per_gpu": head["blocks_per_gpu"], "resident_waves": int(L.LizardGPU_residentWaves()),
                       "size_gather": state["gather"]},
            "ratio": head["ratio"], "compressed_bytes": head["compressed_bytes"],
            "roofline": head["roofline"],
        }
        for k in ("cpu_baseline", "speedup_vs_cpu_1core", "blocks_checked", "blocks_checked_bytes", "checker"):
            if k in head:
                out[k] = head[k]
        if len(results) > 1:
            out["configs"] = results[1:]
        if with_cpu and args.level is None:
            # BASELINE configs[0]: the reference's own CPU-runnable case, and the GPU on exactly that buffer
            c1, hostbuf = cpu_baseline(L, 10, 262144, 256, args.cpu_seconds, seed0=0, whole_buffer=True)
            d = torch.from_numpy(np.frombuffer(hostbuf.raw, dtype=np.uint8).copy()).to(dev)
            _, sz, _ = api.compress_blocks_device(d, 262144, 10)
            torch.cuda.synchronize()
            c1["gpu_compressed_bytes_same_buffer"] = int(sz.to(torch.int64).sum().item())
            c1["gpu_equals_cpu_size"] = c1["gpu_compressed_bytes_same_buffer"] == c1["compressed_bytes"]
            out["config1"] = c1
            # PCIe-inclusive rate of the host-buffer entry on a 4 GiB sample of the headline workload (never `value`)
            nbe = min(16384, head["blocks_per_gpu"]); bs = 262144
            _lib.check(L.LizardGPU_datagen_device(src_all.data_ptr(), nbe, bs, 0.5, 0.0, 0, ctypes.c_void_p(stream.cuda_stream)),
                       "LizardGPU_datagen_device")
            torch.cuda.synchronize()
            host = src_all[:nbe * bs].cpu()
            cap = nbe * api.Lizard_compressBound(bs)
            outbuf = np.empty(cap, dtype=np.uint8)
            e2e = {"sample": f"first {nbe} blocks x {bs} B of the headline workload, LizardGPU_compressBlocks_host_packed, "
                             "256 MiB chunks, three in flight (issuing + draining host threads, pinned staging), device-side compaction, one D2H per chunk",
                   "unit": "MB/s"}
            for name, t in (("pageable_src", host), ("pinned_src", host.pin_memory())):
                best = None
                for _ in range(2):
                    t0 = time.perf_counter()
                    _lib.check(L.LizardGPU_compressBlocks_host_packed(t.data_ptr(), nbe, bs, bs, outbuf.ctypes.data, cap, None, None, 10),
                               "LizardGPU_compressBlocks_host_packed")
                    dt = time.perf_counter() - t0
                    best = dt if best is None else min(best, dt)
                e2e[name] = round(nbe * bs / best / 1e6, 1)
            out["end_to_end"] = e2e
            # .liz frames of the same sample: LizardGPU_compressFrame (block records assembled on the device; XXH32 on a host thread)
            import util
            L.LizardGPU_compressFrameBound.restype = ctypes.c_size_t
            L.LizardGPU_compressFrameBound.argtypes = [ctypes.c_size_t, ctypes.c_void_p]
            L.LizardGPU_compressFrame.restype = ctypes.c_size_t
            L.LizardGPU_compressFrame.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
            fr = {"sample": f"one frame of {nbe * bs} B (256 KiB independent blocks, level 10) from pageable memory, LizardGPU_compressFrame",
                  "unit": "MB/s"}
            for name, crc in (("no_content_checksum", 0), ("with_xxh32_content_checksum", 1)):
                prefs = util.frame_prefs(10, 2, crc, 0)
                fcap = L.LizardGPU_compressFrameBound(nbe * bs, ctypes.byref(prefs))
                fbuf = np.empty(fcap, dtype=np.uint8)
                best = None
                for _ in range(2):
                    t0 = time.perf_counter()
                    n = L.LizardGPU_compressFrame(fbuf.ctypes.data, fcap, host.data_ptr(), nbe * bs, ctypes.byref(prefs))
                    dt = time.perf_counter() - t0
                    assert n < (1 << 63), "LizardGPU_compressFrame failed"
                    best = dt if best is None else min(best, dt)
                fr[name] = round(nbe * bs / best / 1e6, 1)
                fr["frame_bytes"] = int(n)
            out["frames"] = fr
        print(json.dumps(out))
    if world > 1:
        L.LizardGPU_commDestroy()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
