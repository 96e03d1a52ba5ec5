#!/usr/bin/env python3
"""bench.py — Lizard block-compress throughput on MI355X (BASELINE.json metric).

A "step" is one pass of the hot path over one batch: every rank compresses its own blocks of datagen-P50
synthetic input that is already resident in HBM (block b of rank r is RDG_genBuffer(blockSize, 0.5,
seed = r*blocks + b)), then (N > 1) the ranks exchange the per-block compressed sizes with one RCCL all-gather
(inside the library, LizardGPU_gatherSizes_device) so that every rank can compute global output offsets.

  python bench.py --gpus 1 --steps 3 --warmup 1
  python bench.py --gpus N --steps K --warmup W          (no launcher: re-executes itself under torch.distributed.run, one rank per GPU)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  Its top-level fields are BASELINE.json configs[1] (level 10, 65 536 x 256 KiB per
GPU): `value` = whole-job input MB/s (MB = 10^6 B, as reference programs/bench.c:253-255) over the
barrier-bracketed timed region of exactly --steps steps, max over ranks.  The same process then times the other
BASELINE configs with the same --steps / --warmup and reports them under "configs": level 21 and level 30 at
16 384 x 256 KiB (configs[2..3]) and configs[4] in BOTH readings: weak scaling — 6 656 x 4 MiB blocks PER GPU, two per
table-holding wave — and strong scaling — 4 096 x 4 MiB blocks in the WHOLE job (SURVEY 8d read literally), 4 096 / N per GPU, the
same 4 096 blocks at every N ("scaling": "strong", "blocks_total") — the 16 384-block entries also carry "same_inputs_as_headline":
the same kernel on the headline's 65 536 blocks ("same inputs"; kernel time, N = 1) — and, not a BASELINE config, level 20 (fastBig, LIZv1
codewords; round 6) at 16 384 x 256 KiB with the same checks; "config1" is configs[0], the reference's own
CPU-runnable case (64 MiB RDG_genBuffer P50 seed 0 in 256 KiB blocks, programs/bench.c's loop over Lizard_compress).
Per config:
  roofline      achieved = algorithmic bytes per launch (input read once + compressed output written once,
                SURVEY.md §8d) / mean kernel duration (HIP events on the launch stream); traffic = fabric-side bytes
                of one launch from the committed rocprofv3 PMC passes (profiles/pmc_traffic.json)
  cpu_baseline  the stock reference library (oracle/_ref/liblizard_ref.so, kind "reference"; the oracle restatement,
                kind "port", when it is absent) on one host core over a bounded sample of the same blocks
  blocks_checked / blocks_checked_bytes   after the timed region EVERY block's compressed size AND bytes are compared with the
                zero-state reference (oracle/_ref/liblizard_ref_reset.so, else the oracle restatement) run on the host cores
  roofline.traffic_source  which committed counter pass `traffic` comes from (file, round, commit) — `traffic` is null, and this
                says so, when the device sources (lizard_amd/csrc/lz_*.h) no longer hash to what that pass was measured on
  cpu_baseline_all_cores   the same stock reference library in N processes pinned to N host CPUs at once (N stated), each
                looping over its own blocks like programs/bench.c:233-255: the honest per-node CPU figure
"blocks_in_flight" (level 10, 4 MiB blocks): GB/s against the number of blocks of a launch — one wave per block, so a launch
needs about as many blocks as the chip has resident waves before it runs at speed.
N > 1: the size gather is the library's (LizardGPU_gatherSizes_device over RCCL); if it cannot be set up or fails the run FAILS —
there is no torch.distributed fallback — and "per_rank" lists every rank's mean kernel ms and gather us.
N > 1 lines are complete lines: rank 0 still times the CPU reference (`cpu_baseline`, the other ranks wait), EVERY rank verifies
every block of its own shard and `blocks_checked` is the sum over ranks (== n_gpus x blocks per GPU).  --transport host-bounce
(chosen automatically when there are fewer devices than ranks, e.g. two ranks on a one-GPU box) runs torch.distributed on gloo and
installs a transport of the library's size exchange (LizardGPU_setCollectives) that bounces through host memory: the same bench
code path, the same library entry points, no RCCL — `config.size_gather` names the transport; such a line is a functional
check of the N > 1 path, not a scaling measurement (the ranks share a device).
`config.size_gather_transport` (N > 1) is what the transport itself reports (LizardGPU_commInfo): RCCL or an installed table, the
ranks the communicator was made for, the ranks and rank ncclCommCount / ncclCommUserRank report, ncclGetVersion — an RCCL line whose
communicator does not report --gpus ranks is REFUSED, not printed.  `config.scaling_note`: weak scaling, blocks_per_gpu blocks per GPU
at every N (scripts/scale.sh runs N = 1, 2, 4, 8 and prints the efficiencies).
"config1.reference_program": BASELINE configs[0] literally — the reference's own program (oracle/_ref/lizard_cli_ref, `lizard -bL -eL
-B262144 -i3`) on the same 64 MiB buffer, levels 10 / 21 / 30, with the host CPU model.
"one_block_callers": aggregate MB/s of 1..64 host threads calling the reference's one-block Lizard_compress at once (the
combiner, tests/gpu_threads.c).  "concurrent_streams": 4 streams x 20 launches of 64 blocks queued without host synchronisation
(a launch smaller than the machine gets an arena of its own and runs beside the others).  "frames": the reference's own frame entry point LizardF_compressFrame on a 4 GiB host buffer.
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
from tools import datagen as tools_datagen      # bench / test tooling: the synthetic-workload generator (not in the product library)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md

_SIG = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]


def kernel_name(level, bs):
    huf = "true" if level >= 30 else "false"
    base = level - 20 if level >= 30 else level
    if level >= 34 and level <= 38:
        base = level - 21
    if base == 10:
        return "lz_fast12_split_kernel<%s>" % huf
    if base == 11:
        return "lz_fast18_kernel<%s>" % huf
    if base == 20:
        return "lz_fastbig14_kernel<%s>" % huf
    if base == 21:
        return "lz_pricefast14_kernel<%s, %s>" % (huf, "true" if bs <= (256 << 10) else "false")
    if base == 22:
        return "lz_pricefast18_kernel<%s>" % huf
    if level == 32:
        return "lz_hashchain_kernel<true, 6, 14>"
    if level in (12, 33):
        return "lz_hashchain_kernel<%s, 6, 18>" % huf
    return "lz_hashchain_kernel<%s, %d, 18>" % (huf, 7 if base == 13 else 8 if base == 14 else 9 if base == 15 else 4)


def kernel_source_sha16():
    """Hash of the device sources of the COMPRESS kernels (every lz_*.h of lizard_amd/csrc except the decoder lz_unpack.h and the
    host path's compaction lz_pack.h): what a committed counter pass was measured on."""
    import hashlib
    d = os.path.join(ROOT, "lizard_amd", "csrc")
    h = hashlib.sha256()
    for f in sorted(os.listdir(d)):
        if f.startswith("lz_") and f.endswith(".h") and f not in ("lz_unpack.h", "lz_pack.h"):
            h.update(f.encode()); h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def lookup_traffic(level, bs, nb):
    """(traffic bytes per launch | None, traffic_source): the LAST matching entry of profiles/pmc_traffic.json, only if it
    was measured on the device sources of this tree."""
    cur = kernel_source_sha16()
    ent = None
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            for e in json.load(f)["entries"]:
                if (e["level"], e["block_size"], e["blocks_per_gpu"]) == (level, bs, nb):
                    ent = e
    except (OSError, KeyError, ValueError):
        pass
    if ent is None:
        return None, {"file": None, "note": "no counter pass recorded for this configuration", "kernel_source_sha16": cur}
    src = {"file": ent.get("source"), "round": ent.get("round"), "commit": ent.get("commit"),
           "measured_on_kernel_source_sha16": ent.get("kernel_source_sha16"), "kernel_source_sha16": cur,
           "how": "separate rocprofv3 --pmc passes, not measured in this run"}
    if ent.get("kernel_source_sha16") != cur:
        src["stale"] = True
        src["stale_traffic_bytes"] = ent["traffic_bytes"]
        return None, src
    return ent["traffic_bytes"], src


def _all_cores_child(cpu, level, bs, nblocks, seconds, start, out, idx):
    os.sched_setaffinity(0, {cpu})
    import util
    fn, _ = load_checker(zero_state=False)
    buf = ctypes.create_string_buffer(nblocks * bs)
    base = ctypes.addressof(buf)
    for b in range(nblocks):
        util.oracle().lzo_datagen(base + b * bs, bs, 0.5, 0.0, idx * nblocks + b)
    bound = util.oracle().lzo_compress_bound(bs)
    dst = ctypes.create_string_buffer(bound)
    start.wait()
    t0 = time.perf_counter(); done = 0
    while True:                                             # programs/bench.c:233-241: block after block through Lizard_compress
        for b in range(nblocks):
            fn(base + b * bs, dst, bs, bound, level)
            done += 1
        if time.perf_counter() - t0 >= seconds:
            break
    out[2 * idx] = done * bs
    out[2 * idx + 1] = time.perf_counter() - t0


def cpu_quota_cores():
    """CPU time this container may use, in cores (cgroup v2 cpu.max / v1 cfs quota), or None when unlimited."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        return None if q == "max" else float(q) / float(per)
    except (OSError, ValueError):
        pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else q / per
    except (OSError, ValueError):
        return None


def cpu_all_cores(level, bs, nblocks, seconds):
    """N processes pinned one per CPU, all compressing at once; whole-node MB/s.  N = the CPUs this process may run on, capped
    by the container's CPU quota when there is one (more runnable processes than quota only adds throttling)."""
    import multiprocessing as mp
    visible = sorted(os.sched_getaffinity(0))
    quota = cpu_quota_cores()
    cpus = visible if quota is None else visible[:max(1, min(len(visible), int(quota + 0.5)))]
    if os.environ.get("LIZARD_BENCH_NPROC"):                 # a second point of the curve (bench.py asks for 8 next to "all")
        cpus = cpus[:max(1, int(os.environ["LIZARD_BENCH_NPROC"]))]
    ctx = mp.get_context("fork")
    start = ctx.Barrier(len(cpus) + 1)
    out = ctx.Array("d", 2 * len(cpus), lock=False)
    procs = [ctx.Process(target=_all_cores_child, args=(c, level, bs, nblocks, seconds, start, out, i)) for i, c in enumerate(cpus)]
    for p in procs: p.start()
    start.wait()
    for p in procs: p.join()
    if any(p.exitcode for p in procs):
        raise SystemExit("cpu_all_cores: a worker failed")
    tot = sum(out[2 * i] for i in range(len(cpus)))
    tmax = max(out[2 * i + 1] for i in range(len(cpus)))
    _, kind = load_checker(zero_state=False)
    return {"value": round(tot / tmax / 1e6, 1), "unit": "MB/s", "cores": len(cpus), "kind": kind,
            "sample": f"{len(cpus)} processes pinned one per logical CPU, each looping over its own {nblocks} blocks x {bs} B "
                      f"(datagen P50) through Lizard_compress level {level} for >= {seconds} s, all at once",
            "per_process_mb_s": round(tot / tmax / 1e6 / len(cpus), 1), "host_cpus_visible": len(visible),
            "container_cpu_quota_cores": quota}


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
        tools_datagen.datagen_host(base, n_blocks * block_size, 0.5, 0.0, seed0)
    else:
        for b in range(n_blocks):
            tools_datagen.datagen_host(base + b * block_size, block_size, 0.5, 0.0, seed0 + b)
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
            "ratio": round(nbytes / total_c, 4), "compressed_bytes": total_c, "host_cpus": os.cpu_count(), "host_cpu": host_cpu_model()}, buf


def host_cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return None


def reference_program_bench(hostbuf, levels, seconds=3):
    """BASELINE configs[0] literally: the reference's OWN program — oracle/_ref/lizard_cli_ref, its
    programs/*.c compiled against its own lib/ by oracle/Makefile — in benchmark mode on the same 64 MiB buffer the ctypes loop
    above timed (programs/bench.c:231-255: blocks of -B bytes through Lizard_compress one after the other, fastest loop of -i
    seconds, MB = 10^6 B, :321-328 the result line).  Absent binary (it is git-ignored and built where /root/reference exists):
    says so.  One thread, like every run of the reference."""
    import re
    import subprocess
    import tempfile
    exe = os.path.join(ROOT, "oracle", "_ref", "lizard_cli_ref")
    if not os.path.exists(exe):
        return {"note": "oracle/_ref/lizard_cli_ref not present (built by oracle/Makefile where /root/reference exists)"}
    res = {"program": "oracle/_ref/lizard_cli_ref (the reference's programs/ + lib/, unmodified, -O3)", "host_cpu": host_cpu_model(),
           "host_cpus": os.cpu_count(), "threads": 1, "levels": []}
    with tempfile.NamedTemporaryFile(prefix="lizard_bench_p50_", suffix=".bin", dir="/tmp", delete=False) as f:
        f.write(hostbuf.raw)
        path = f.name
    try:
        for lv in levels:
            cmd = [exe, f"-b{lv}", f"-e{lv}", "-B262144", f"-i{seconds}", "-q", path]
            try:
                r = subprocess.run(cmd, capture_output=True, text=True, timeout=120)
            except subprocess.TimeoutExpired:
                res["levels"].append({"level": lv, "error": "timeout"})
                continue
            # "-10    40533852 (1.656) 690.12 MB/s 2873.8 MB/s  name"
            m = re.search(r"-%d\s+(\d+)\s+\(([\d.]+)\)\s+([\d.]+) MB/s\s+([\d.]+) MB/s" % lv, (r.stdout + r.stderr).replace("\r", "\n"))
            if not m:
                res["levels"].append({"level": lv, "error": "could not parse the program's output", "output": (r.stdout + r.stderr)[-200:]})
                continue
            res["levels"].append({"level": lv, "command": " ".join(cmd[:-1]) + " <64 MiB RDG_genBuffer P50 seed 0>", "compressed_bytes": int(m.group(1)),
                                  "ratio": float(m.group(2)), "compress_MB_s": float(m.group(3)), "decompress_MB_s": float(m.group(4))})
    finally:
        os.unlink(path)
    return res


def verify_all_blocks(L, level, bs, nb, src, dst, sizes, stride, seed0, byte_blocks, threads):
    """Checker leg: EVERY block's size and bytes of the last launch against the zero-state reference run on the host cores
    (`byte_blocks` > 0; the device output comes over in chunks, one D2H per chunk).  Also pins the device generator to the host
    generator on a sample."""
    import numpy as np
    fn, kind = load_checker(zero_state=True)
    import util
    bound = util.oracle().lzo_compress_bound(bs)
    sz = sizes.cpu().numpy().astype(np.int64)
    chunk = max(1, min(nb, (512 << 20) // bs))
    tls = {}
    n_bytes_checked = 0
    _memcmp = ctypes.CDLL(None).memcmp
    _memcmp.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]

    def one(args):
        ptr, b, outp = args
        import threading
        t = threading.get_ident()
        if t not in tls:
            tls[t] = ctypes.create_string_buffer(bound)
        out = tls[t]
        r = fn(ptr, out, bs, bound, level)
        same = r == int(sz[b]) and _memcmp(outp, out, r) == 0
        return b, r, same

    with concurrent.futures.ThreadPoolExecutor(max_workers=threads) as pool:
        for c0 in range(0, nb, chunk):
            c1 = min(nb, c0 + chunk)
            host = src[c0 * bs:c1 * bs].cpu().numpy()
            hout = dst[c0 * stride:c1 * stride].cpu().numpy()              # this chunk's output slots
            base, obase = host.ctypes.data, hout.ctypes.data
            for b, r, same in pool.map(one, [(base + (b - c0) * bs, b, obase + (b - c0) * stride) for b in range(c0, c1)]):
                assert r == int(sz[b]), f"level {level}: block {b}: GPU size {int(sz[b])} != reference size {r}"
                assert same, f"level {level}: block {b}: GPU bytes differ from the reference"
                n_bytes_checked += 1
            if c0 == 0:                                     # device datagen == host datagen (what the workload claims to be)
                blk = ctypes.create_string_buffer(bs)
                for b in (0, 1, c1 - 1):
                    tools_datagen.datagen_host(blk, bs, 0.5, 0.0, seed0 + b)
                    assert host[(b - c0) * bs:(b - c0 + 1) * bs].tobytes() == blk.raw, f"device datagen differs from host datagen at block {b}"
    return nb, n_bytes_checked, kind


def launcher_argv(n, argv):
    """The command `python bench.py --gpus N ...` turns itself into when no launcher started it: the driver's own N > 1 form."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--level", type=int, default=None, help="time ONE configuration only (with --block-size / --blocks)")
    ap.add_argument("--block-size", type=int, default=262144)
    ap.add_argument("--blocks", type=int, default=None, help="blocks per GPU (weak scaling)")
    ap.add_argument("--verify", type=int, default=1, help="1: every block of every config is compared with the reference, sizes and bytes; 0 = no check")
    ap.add_argument("--cpu-seconds", type=float, default=5.0, help="CPU baseline budget per config")
    ap.add_argument("--cpu-blocks", type=int, default=256)
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline / verification / end-to-end legs")
    ap.add_argument("--headline-only", action="store_true")
    ap.add_argument("--strong", action="store_true", help="with --headline-only: also time the strong-scaling reading of BASELINE configs[4]")
    ap.add_argument("--strong-blocks", type=int, default=4096, help="4 MiB blocks in the WHOLE job of the strong-scaling entry (divided by --gpus)")
    ap.add_argument("--transport", choices=("auto", "rccl", "host-bounce"), default="auto",
                    help="N > 1: the size exchange over RCCL (nccl backend), or bounced through host memory over gloo (ranks may share a device)")
    ap.add_argument("--cpu-all-seconds", type=float, default=3.0, help="all-cores CPU baseline: seconds per level (0 = skip)")
    ap.add_argument("--cpu-all-cores-worker", nargs=4, type=float, default=None, metavar=("LEVEL", "BS", "NBLOCKS", "SECONDS"),
                    help="internal: run the all-cores CPU baseline in this (GPU-free) process and print its JSON")
    args = ap.parse_args()
    if args.cpu_all_cores_worker:
        lv, bs_, nbk, sec = args.cpu_all_cores_worker
        print(json.dumps(cpu_all_cores(int(lv), int(bs_), int(nbk), sec)))
        return

    import numpy as np
    import torch
    import torch.distributed as dist
    from lizard_amd import _lib, api

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` on its own: become the launcher — one rank per GPU under torch.distributed.run, same argv,
        # a free port on 127.0.0.1 (the container's hostname may not resolve).  The ranks' single JSON line is this process's.
        cmd = launcher_argv(args.gpus, sys.argv[1:])
        sys.stdout.flush(); sys.stderr.flush()
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.execv(cmd[0], cmd)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher's world size and --gpus disagree")
    ndev = torch.cuda.device_count()
    transport = args.transport
    if transport == "auto":
        transport = "host-bounce" if (world > 1 and ndev < world) else "rccl"
    dev_index = local_rank % max(1, ndev) if transport == "host-bounce" else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    cdev = dev if transport == "rccl" else torch.device("cpu")      # where torch.distributed's small control tensors live
    if world > 1:
        if transport == "rccl":
            dist.init_process_group("nccl", device_id=dev)   # "nccl" is RCCL on ROCm
        else:
            dist.init_process_group("gloo")

    L = _lib.lib()
    _lib.check(L.LizardGPU_setDevice(dev_index), "LizardGPU_setDevice")

    # the size gather lives in the library; the launcher's transport only carries the 128-byte id (RCCL) or is handed to the
    # library as its collective table (host-bounce).  No fallback: a job whose library-side exchange cannot be set up fails here.
    gather_via = "none (1 GPU)"
    keep_alive = []
    if world > 1 and transport == "rccl":
        uid = torch.zeros(128, dtype=torch.uint8, device=dev)
        if rank == 0:
            buf = ctypes.create_string_buffer(128)
            _lib.check(L.LizardGPU_commUniqueId(buf), "LizardGPU_commUniqueId")
            uid.copy_(torch.frombuffer(bytearray(buf.raw), dtype=torch.uint8))
        dist.broadcast(uid, 0)
        _lib.check(L.LizardGPU_commInitRank(bytes(uid.cpu().numpy().tobytes()), world, rank), "LizardGPU_commInitRank")
        L.LizardGPU_rcclShared.restype = ctypes.c_int
        gather_via = ("library: ncclAllGather via LizardGPU_gatherSizes_device (RCCL "
                      + ("shared with torch" if L.LizardGPU_rcclShared() == 1 else "loaded by the library") + ")")
    elif world > 1:
        hip = ctypes.CDLL("libamdhip64.so")
        hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
        hip.hipStreamSynchronize.argtypes = [ctypes.c_void_p]

        def _d2h(ptr, count):
            t = torch.empty(count, dtype=torch.int32)
            assert hip.hipMemcpy(t.data_ptr(), ptr, count * 4, 2) == 0
            return t

        def _h2d(ptr, t):
            assert hip.hipMemcpy(ptr, t.data_ptr(), t.numel() * 4, 1) == 0

        def _all_gather(send, recv, count, comm, stream_):
            try:
                hip.hipStreamSynchronize(stream_)
                mine = _d2h(send, count)
                outs = [torch.empty(count, dtype=torch.int32) for _ in range(world)]
                dist.all_gather(outs, mine)
                _h2d(recv, torch.cat(outs))
                return 0
            except Exception as e:                            # noqa: BLE001 - a Python exception must not unwind through C
                sys.stderr.write(f"host-bounce allGather failed: {e}\n")
                return -6

        def _broadcast(send, recv, count, root, comm, stream_):
            try:
                hip.hipStreamSynchronize(stream_)
                t = _d2h(send, count) if rank == root else torch.empty(count, dtype=torch.int32)
                dist.broadcast(t, root)
                _h2d(recv, t)
                return 0
            except Exception as e:                            # noqa: BLE001
                sys.stderr.write(f"host-bounce broadcast failed: {e}\n")
                return -6

        AG = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p)
        BC = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p)
        GR = ctypes.CFUNCTYPE(ctypes.c_int)

        class Table(ctypes.Structure):
            _fields_ = [("allGather", AG), ("broadcast", BC), ("groupStart", GR), ("groupEnd", GR)]
        table = Table(AG(_all_gather), BC(_broadcast), GR(lambda: 0), GR(lambda: 0))
        keep_alive.append(table)
        L.LizardGPU_setCollectives.argtypes = [ctypes.c_void_p]
        _lib.check(L.LizardGPU_setCollectives(ctypes.byref(table)), "LizardGPU_setCollectives")
        _lib.check(L.LizardGPU_commInitRank(bytes(128), world, rank), "LizardGPU_commInitRank")
        gather_via = (f"library: LizardGPU_gatherSizes_device over a transport installed with LizardGPU_setCollectives — "
                      f"host bounce (hipMemcpy + torch.distributed gloo), {world} ranks on {ndev} device(s); functional check, not a scaling measurement")
    comm_info = None
    if world > 1:
        info = (ctypes.c_int * 6)()
        L.LizardGPU_commInfo(info)
        comm_info = {"transport": "rccl" if info[0] == 0 else "LizardGPU_setCollectives table", "ranks_requested": info[1],
                     "rccl_ranks_seen": info[2], "rccl_rank_seen": info[3], "rccl_version": info[4]}
        # the line must not quote an N-GPU figure unless the transport itself saw N ranks
        if transport == "rccl" and info[2] != world:
            raise SystemExit(f"bench.py rank {rank}: --gpus {world} but the library's RCCL communicator reports {info[2]} ranks")
        if info[1] != world:
            raise SystemExit(f"bench.py rank {rank}: --gpus {world} but the library's rank communicator was made for {info[1]} ranks")

    # (level, block size, blocks PER GPU, scaling): "weak" = the same blocks per GPU at every N; "strong" = BASELINE configs[4] read
    # literally — 4 096 x 4 MiB frame blocks in the whole job, 4 096 / N per GPU, the same 4 096 blocks at every N
    if args.level is not None:
        plan = [(args.level, args.block_size, args.blocks or 16384, "weak")]
    else:
        plan = [(10, 262144, args.blocks or 65536, "weak")]
        if not args.headline_only:
            plan += [(21, 262144, 16384, "weak"), (30, 262144, 16384, "weak"), (10, 4 << 20, 6656, "weak")]
            plan += [(20, 262144, 16384, "weak")]          # not a BASELINE config: the fastBig level of round 6 (LIZv1 codewords), same checks
        if not args.headline_only or args.strong:
            plan += [(10, 4 << 20, max(1, args.strong_blocks // world), "strong")]
    max_in = max(nb * bs for _, bs, nb, _ in plan)
    max_out = max(nb * ((api.Lizard_compressBound(bs) + 63) & ~63) for _, bs, nb, _ in plan)
    src_all = torch.empty(max_in, dtype=torch.uint8, device=dev)
    dst_all = torch.empty(max_out, dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream(dev)
    threads = max(1, min(96, int(cpu_quota_cores() or ((os.cpu_count() or 8) - 2))))      # checker threads: the CPUs this container may use
    state = {"gather": gather_via}

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def run_config(level, bs, nb, with_cpu, scaling="weak"):
        if not L.LizardGPU_levelSupported(level):
            raise SystemExit(f"level {level} is not implemented on the GPU path")
        stride = (api.Lizard_compressBound(bs) + 63) & ~63
        src, dst = src_all[:nb * bs], dst_all[:nb * stride]
        sizes = torch.zeros(nb, dtype=torch.int32, device=dev)
        all_sizes = torch.zeros(world * nb, dtype=torch.int32, device=dev)
        offsets = torch.zeros(world * nb + 1, dtype=torch.int64, device=dev)
        seed0 = rank * nb
        tools_datagen.datagen_device(src.data_ptr(), nb, bs, 0.5, 0.0, seed0, ctypes.c_void_p(stream.cuda_stream))
        torch.cuda.synchronize()
        kernel_ms = []
        gathered = [None]

        gather_ev = []

        def step(timed):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            api.compress_blocks_device(src, bs, level, dst=dst, sizes=sizes)
            e1.record(stream)
            if world > 1:                                     # RCCL over xGMI: 4 B per block per rank
                rc = L.LizardGPU_gatherSizes_device(sizes.data_ptr(), world * nb, all_sizes.data_ptr(), offsets.data_ptr(),
                                                    ctypes.c_void_p(stream.cuda_stream))
                if rc != 0:
                    raise SystemExit(f"bench.py rank {rank}: LizardGPU_gatherSizes_device failed ({rc}): "
                                     + L.LizardGPU_lastError().decode(errors="replace"))
                gathered[0] = (all_sizes, offsets[:-1])
                if timed:
                    e2 = torch.cuda.Event(enable_timing=True); e2.record(stream)
                    gather_ev.append((e1, e2))
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
            t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        kms = [a.elapsed_time(b) for a, b in kernel_ms]
        per_rank = None
        if world > 1:                                        # every rank's mean kernel ms and gather us, gathered for the line
            mine = torch.tensor([sum(kms) / len(kms), 1e3 * sum(a.elapsed_time(b) for a, b in gather_ev) / max(1, len(gather_ev))],
                                dtype=torch.float64, device=cdev)
            allr = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(allr, mine)
            per_rank = [{"rank": r, "kernel_ms": round(float(t[0]), 3), "gather_us": round(float(t[1]), 1)} for r, t in enumerate(allr)]
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
            traffic, traffic_source = lookup_traffic(level, bs, nb)   # fabric bytes per launch from a committed counter pass of THESE kernels, else null
            res = {
                "level": level, "block_size": bs, "blocks_per_gpu": nb, "blocks_total": nb * world, "scaling": scaling,
                "workload": f"L{level} {nb}x{bs}B blocks/GPU x {world} GPU ({scaling}), datagen P50 seed=block index, HBM-resident",
                "value": round(tot_in * args.steps / elapsed / 1e6, 1), "unit": "MB/s",
                "ms_per_step": round(elapsed / args.steps * 1e3, 3),
                "ratio": round(tot_in / tot_out, 4), "compressed_bytes": tot_out,
                "roofline": {"bound": "hbm", "achieved": round(alg_bytes / avg_k / 1e9, 2), "peak": HBM_PEAK_GBS,
                             "unit": "GB/s", "frac": round(alg_bytes / avg_k / 1e9 / HBM_PEAK_GBS, 5), "traffic": traffic,
                             "traffic_source": traffic_source,
                             "traffic_over_algorithmic": round(traffic / alg_bytes, 2) if traffic else None,
                             "kernel": kernel_name(level, bs), "avg_kernel_ms": round(avg_k * 1e3, 3),
                             "algorithmic_bytes_per_launch": alg_bytes},
            }
            if per_rank:
                res["per_rank"] = per_rank
            if with_cpu:                            # the CPU legs: the only place the oracle / oracle/_ref is touched
                ncpu = min(args.cpu_blocks if bs <= (1 << 20) else 32, nb)
                res["cpu_baseline"], _ = cpu_baseline(L, level, bs, ncpu, args.cpu_seconds)
                res["speedup_vs_cpu_1core"] = round(res["value"] / res["cpu_baseline"]["value"], 2)
                if args.cpu_all_seconds > 0 and bs <= (1 << 20) and world == 1:
                    import subprocess
                    w = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-all-cores-worker", str(level), str(bs), "32", str(args.cpu_all_seconds)],
                                       capture_output=True, text=True, timeout=300)
                    if w.returncode == 0:
                        res["cpu_baseline_all_cores"] = json.loads(w.stdout.strip().splitlines()[-1])
                        res["speedup_vs_cpu_all_cores"] = round(res["value"] / res["cpu_baseline_all_cores"]["value"], 2)
                        if res["cpu_baseline_all_cores"]["cores"] > 8:      # does the box give this container all of those CPUs? 8 processes beside it
                            w8 = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-all-cores-worker", str(level), str(bs), "32", "1.5"],
                                                capture_output=True, text=True, timeout=300, env=dict(os.environ, LIZARD_BENCH_NPROC="8"))
                            if w8.returncode == 0:
                                r8 = json.loads(w8.stdout.strip().splitlines()[-1])
                                res["cpu_baseline_all_cores"]["with_8_processes"] = {"value": r8["value"], "per_process_mb_s": r8["per_process_mb_s"]}
                    else:
                        res["cpu_baseline_all_cores"] = {"error": w.stderr[-300:]}
        # every rank checks EVERY block of its own shard (sizes and bytes) against the zero-state reference; rank 0's CPU timing
        # above runs while the others wait, so that it is not disturbed by their checker threads
        if with_cpu and world > 1:
            dist.barrier()
        if with_cpu and args.verify > 0:
            t0 = time.perf_counter()
            n_sz, n_by, kind = verify_all_blocks(L, level, bs, nb, src, dst, sizes, stride, seed0, args.verify, max(1, threads // world))
            if world > 1:
                t = torch.tensor([n_sz, n_by], dtype=torch.int64, device=cdev)
                dist.all_reduce(t, op=dist.ReduceOp.SUM)
                n_sz, n_by = int(t[0].item()), int(t[1].item())
            if rank == 0:
                res["blocks_checked"] = n_sz
                res["blocks_checked_bytes"] = n_by
                res["checker"] = (f"{kind} (zero-state), {max(1, threads // world)} host threads per rank, every rank its own shard, "
                                  f"{round(time.perf_counter() - t0, 1)} s")
        return res, (src, dst, sizes, stride)

    with_cpu = not args.no_cpu
    results = []
    for level, bs, nb, scaling in plan:
        r, _ = run_config(level, bs, nb, with_cpu, scaling)
        results.append(r)

    # SURVEY 8d gives configs[2..3] (levels 21 / 30) "the same inputs" as configs[1]; the line has carried them at 16 384 blocks per GPU
    # since round 1 (5 - 6 blocks per block-claiming wave: the end of the launch, where waves run out of blocks one by one, is a
    # visible part of it).  The same kernels on the headline's 65 536 blocks, kernel time over 3 launches, ride beside each of them.
    if world == 1 and args.level is None and args.blocks is None and not args.headline_only:
        nb_h, bs_h = 65536, 262144
        for r in results[1:] if rank == 0 else []:
            if r["block_size"] != bs_h or r["blocks_per_gpu"] >= nb_h or r["scaling"] != "weak":
                continue
            stride = (api.Lizard_compressBound(bs_h) + 63) & ~63
            src, dst = src_all[:nb_h * bs_h], dst_all[:nb_h * stride]
            sizes = torch.zeros(nb_h, dtype=torch.int32, device=dev)
            tools_datagen.datagen_device(src.data_ptr(), nb_h, bs_h, 0.5, 0.0, 0, ctypes.c_void_p(stream.cuda_stream))
            ev = []
            for i in range(4):
                e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                e0.record(stream); api.compress_blocks_device(src, bs_h, r["level"], dst=dst, sizes=sizes); e1.record(stream)
                if i:
                    ev.append((e0, e1))
            torch.cuda.synchronize()
            ms = sum(a.elapsed_time(b) for a, b in ev) / len(ev)
            r["same_inputs_as_headline"] = {"blocks_per_gpu": nb_h, "value": round(nb_h * bs_h / ms / 1e3, 1), "unit": "MB/s (kernel time)",
                                            "avg_kernel_ms": round(ms, 3), "launches": len(ev),
                                            "compressed_bytes": int(sizes.to(torch.int64).sum().item())}

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
                       "size_gather": state["gather"], "size_gather_transport": comm_info,
                       "scaling_note": ("top level and configs[] with scaling=weak: every GPU compresses blocks_per_gpu blocks of its own. "
                                        "BASELINE configs[4] (4 MiB frame blocks over 8 GPUs) is carried twice: weak, 6 656 blocks PER GPU "
                                        "(two per table-holding wave; a launch needs >= 3 328 blocks to fill one MI355X), and strong, "
                                        "4 096 blocks in the WHOLE job = 4 096 / N per GPU (SURVEY 8d read literally)")},
            "ratio": head["ratio"], "compressed_bytes": head["compressed_bytes"],
            "roofline": head["roofline"],
        }
        for k in ("cpu_baseline", "speedup_vs_cpu_1core", "cpu_baseline_all_cores", "speedup_vs_cpu_all_cores", "per_rank",
                  "blocks_checked", "blocks_checked_bytes", "checker"):
            if k in head:
                out[k] = head[k]
        if len(results) > 1:
            out["configs"] = results[1:]
        if world == 1 and args.level is None and not args.headline_only:
            # one wave = one block: what a launch of fewer blocks than resident waves reaches (level 10, 4 MiB blocks; src_all still holds them)
            bs4 = 4 << 20
            stride4 = (api.Lizard_compressBound(bs4) + 63) & ~63
            curve = []
            for nbk in (256, 512, 1024, 3328, 6656):
                sz4 = torch.zeros(nbk, dtype=torch.int32, device=dev)
                ms = []
                for it in range(3):
                    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                    e0.record(stream)
                    api.compress_blocks_device(src_all[:nbk * bs4], bs4, 10, dst=dst_all[:nbk * stride4], sizes=sz4)
                    e1.record(stream)
                    torch.cuda.synchronize()
                    if it:
                        ms.append(e0.elapsed_time(e1))
                curve.append({"blocks": nbk, "GB_s": round(nbk * bs4 / (sum(ms) / len(ms)) / 1e6, 1)})
            out["blocks_in_flight"] = {"workload": "level -10, 4 MiB blocks (the CLI's default block size), datagen P50, device-resident, kernel time",
                                       "resident_waves": int(L.LizardGPU_residentWaves()), "curve": curve}
        if with_cpu and world == 1 and args.level is None:
            # BASELINE configs[0]: the reference's own CPU-runnable case, and the GPU on exactly that buffer — 256 blocks, a launch
            # far smaller than the machine (one block per CU): kernel time device-resident, and the PCIe-inclusive host-buffer call
            c1, hostbuf = cpu_baseline(L, 10, 262144, 256, args.cpu_seconds, seed0=0, whole_buffer=True)
            d = torch.from_numpy(np.frombuffer(hostbuf.raw, dtype=np.uint8).copy()).to(dev)
            kms1 = []
            for it in range(4):
                e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                _, sz, _ = api.compress_blocks_device(d, 262144, 10)
                e1.record(stream)
                torch.cuda.synchronize()
                if it:
                    kms1.append(e0.elapsed_time(e1))
            c1["gpu_compressed_bytes_same_buffer"] = int(sz.to(torch.int64).sum().item())
            c1["gpu_equals_cpu_size"] = c1["gpu_compressed_bytes_same_buffer"] == c1["compressed_bytes"]
            c1["gpu_kernel_ms_same_buffer"] = round(min(kms1), 3)
            c1["gpu_MB_s_same_buffer_device_resident"] = round(256 * 262144 / min(kms1) / 1e3, 1)
            hb = np.frombuffer(hostbuf.raw, dtype=np.uint8)
            ob = np.empty(256 * api.Lizard_compressBound(262144), dtype=np.uint8)
            best = None
            for _ in range(3):
                t0 = time.perf_counter()
                _lib.check(L.LizardGPU_compressBlocks_host_packed(hb.ctypes.data, 256, 262144, 262144, ob.ctypes.data, ob.size, None, None, 10),
                           "LizardGPU_compressBlocks_host_packed")
                dt = time.perf_counter() - t0
                best = dt if best is None else min(best, dt)
            c1["gpu_MB_s_same_buffer_host_to_host"] = round(256 * 262144 / best / 1e6, 1)
            c1["reference_program"] = reference_program_bench(hostbuf, (10, 21, 30))
            out["config1"] = c1
            # N host threads in the reference's ONE-BLOCK entry point at once (the combiner): tests/gpu_threads.c, every call checked
            exe = os.path.join(ROOT, "tests", "gpu_threads")
            if os.path.exists(exe):
                import subprocess
                w = subprocess.run([exe, "64", "10", "262144", "1.5", "json"], capture_output=True, text=True, timeout=120)
                if w.returncode == 0:
                    out["one_block_callers"] = {
                        "sample": "N host threads, each calling Lizard_compress(256 KiB datagen P50 block, level 10) in a loop for 1.5 s, "
                                  "host buffers, every result compared with the oracle; callers that arrive while a launch is in flight "
                                  "leave together in the next one (one ragged batch, one block per CU)",
                        "unit": "MB/s aggregate", "curve": json.loads(w.stdout.strip().splitlines()[-1])}
                else:
                    out["one_block_callers"] = {"error": (w.stdout + w.stderr)[-300:]}
            # small launches on several streams at once (arenas, include/lizard_amd.h LizardGPU_arenasInUse): 4 streams x 20 launches of
            # 64 blocks each, queued without host synchronisation; wall time from the first launch to the last one's end
            try:
                S, R, nbs, bss = 4, 20, 64, 262144
                sstreams = [torch.cuda.Stream(device=dev) for _ in range(S)]
                ssrc = src_all[:nbs * bss]
                sbufs = [api.compress_blocks_device(ssrc, bss, 10) for _ in range(S)]
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(R):
                    for k in range(S):
                        with torch.cuda.stream(sstreams[k]):
                            api.compress_blocks_device(ssrc, bss, 10, dst=sbufs[k][0], sizes=sbufs[k][1])
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                same = all(torch.equal(sbufs[0][1], sbufs[k][1]) for k in range(1, S))
                L.LizardGPU_arenasInUse.restype = ctypes.c_int
                out["concurrent_streams"] = {"sample": f"{S} streams x {R} launches of {nbs} x {bss} B blocks at level -10, device-resident, no host "
                                                       "synchronisation in between; a launch smaller than the machine gets an arena of its own",
                                             "arenas_in_use": int(L.LizardGPU_arenasInUse()), "MB_s": round(S * R * nbs * bss / dt / 1e6, 1),
                                             "all_streams_same_sizes": bool(same)}
            except Exception as ex:                               # noqa: BLE001 — a side measurement must not take the line down
                out["concurrent_streams"] = {"error": repr(ex)[:200]}
            # PCIe-inclusive rate of the host-buffer entry on a 4 GiB sample of the headline workload (never `value`)
            nbe = min(16384, head["blocks_per_gpu"]); bs = 262144
            tools_datagen.datagen_device(src_all.data_ptr(), nbe, bs, 0.5, 0.0, 0, ctypes.c_void_p(stream.cuda_stream))
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
            # .liz frames of the same sample through the REFERENCE'S OWN frame entry point, LizardF_compressFrame (lib/lizard_frame.h):
            # every block of the call is one batch (block records assembled on the device; XXH32 on a host thread)
            import util
            L.LizardF_compressFrameBound.restype = ctypes.c_size_t
            L.LizardF_compressFrameBound.argtypes = [ctypes.c_size_t, ctypes.c_void_p]
            L.LizardF_compressFrame.restype = ctypes.c_size_t
            L.LizardF_compressFrame.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
            fr = {"sample": f"one frame of {nbe * bs} B from pageable memory through LizardF_compressFrame, level 10: 256 KiB independent blocks "
                            "without / with the content checksum, and with NULL preferences (the reference's defaults: 128 KiB linked blocks, "
                            "level 17 — linked frames carry independently compressed blocks here)",
                  "unit": "MB/s"}
            for name, prefs in (("no_content_checksum", util.frame_prefs(10, 2, 0, 0)), ("with_xxh32_content_checksum", util.frame_prefs(10, 2, 1, 0)),
                                ("null_preferences", None)):
                pref_ptr = ctypes.byref(prefs) if prefs is not None else None
                fcap = L.LizardF_compressFrameBound(nbe * bs, pref_ptr)
                fbuf = np.empty(fcap, dtype=np.uint8)
                best = None
                for _ in range(2):
                    t0 = time.perf_counter()
                    n = L.LizardF_compressFrame(fbuf.ctypes.data, fcap, host.data_ptr(), nbe * bs, pref_ptr)
                    dt = time.perf_counter() - t0
                    assert n < (1 << 63), "LizardF_compressFrame failed"
                    best = dt if best is None else min(best, dt)
                fr[name] = round(nbe * bs / best / 1e6, 1)
                fr["frame_bytes_" + name] = int(n)
            out["frames"] = fr
        print(json.dumps(out))
    if world > 1:
        L.LizardGPU_commDestroy()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
