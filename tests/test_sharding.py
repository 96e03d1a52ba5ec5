"""CPU, world_size 2 over gloo: the N>1 path (block partition, size all-gather, global offsets).
The per-block compressor in this test is the oracle (checker); on the GPU box the same code runs with
the HIP path and RCCL (bench.py --gpus N)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import util
from torch_sharding import gather_block_sizes, shard_range


def test_shard_range_is_a_partition():
    for n in (1, 2, 7, 8, 65536, 65537):
        for world in (1, 2, 3, 8):
            covered = []
            for r in range(world):
                s, c = shard_range(n, r, world)
                covered += list(range(s, s + c))
            assert covered == list(range(n))


def _worker(rank, world, port, n_blocks, bs, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    start, count = shard_range(n_blocks, rank, world)
    outs = [util.oracle_compress(util.datagen(bs, 0.5, 0.0, start + i), 10) for i in range(count)]
    local = torch.tensor([len(o) for o in outs], dtype=torch.int32)
    sizes, offsets = gather_block_sizes(local, n_blocks)
    q.put((rank, sizes.tolist(), offsets.tolist(), [(start + i, util.sha(o)) for i, o in enumerate(outs)]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_gather_sizes_and_offsets():
    n_blocks, bs, world = 7, 65536, 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_blocks, bs, q)) for r in range(world)]
    for p in procs: p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs: p.join(60)
    want = [len(util.oracle_compress(util.datagen(bs, 0.5, 0.0, b), 10)) for b in range(n_blocks)]
    want_off = list(np.cumsum([0] + want[:-1]))
    seen = {}
    for rank, sizes, offsets, hashes in res:
        assert sizes == want and offsets == want_off, rank
        seen.update(dict(hashes))
    assert sorted(seen) == list(range(n_blocks))


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus N` without a launcher: the command it execs is the driver's own N > 1 form, and
    on this GPU-less box the exec really happens — N ranks start under torch.distributed.run and each one says there is no device."""
    import subprocess
    import sys
    sys.path.insert(0, util.ROOT)
    import bench
    cmd = bench.launcher_argv(4, ["--gpus", "4", "--steps", "2"])
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    assert cmd[-5:] == [os.path.join(util.ROOT, "bench.py"), "--gpus", "4", "--steps", "2"]
    if torch.cuda.is_available():
        return
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(util.ROOT, "bench.py"), "--gpus", "2", "--headline-only", "--no-cpu", "--blocks", "8"],
                       capture_output=True, text=True, timeout=600, env=env, cwd=util.ROOT)
    assert r.returncode != 0
    err = r.stdout + r.stderr
    assert "launch with torch.distributed.run" not in err and "disagree" not in err
    assert "local_rank: 1" in err or "rank      : 1" in err or "rank: 1" in err.replace(" ", "").replace("rank:", "rank: "), err[-1500:]
