"""GPU: the N > 1 branch of bench.py executed for real on a one-GPU box: two torchrun ranks share the
device; torch.distributed runs on gloo and the library's size exchange (LizardGPU_commInitRank + LizardGPU_gatherSizes_device)
goes through a host-bounce transport installed with LizardGPU_setCollectives.  Same bench code path as an RCCL job: per-rank
kernel / gather timings, rank 0's cpu_baseline, every rank verifying every block of its shard."""
import json
import os
import socket
import subprocess
import sys

import pytest

import util


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.gpu
@pytest.mark.parametrize("world,blocks,self_launch", [(2, 1024, True), (3, 683, False)])
def test_bench_two_and_three_ranks_on_one_device(world, blocks, self_launch):
    """self_launch: `python bench.py --gpus N` with no launcher on the command line and no WORLD_SIZE in the environment — the form
    the driver uses for N = 1 — must start its own ranks; the other case is the driver's torchrun form."""
    import torch
    if torch.cuda.device_count() >= world:
        extra = ["--transport", "host-bounce"]               # (a box with enough devices: still exercise the bounce transport here)
    else:
        extra = []
    tail = ["--gpus", str(world), "--headline-only", "--strong", "--strong-blocks", "96",
            "--blocks", str(blocks), "--steps", "2", "--warmup", "1", "--cpu-seconds", "1", "--cpu-all-seconds", "0"] + extra
    if self_launch:
        cmd = [sys.executable, os.path.join(util.ROOT, "bench.py")] + tail
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port()), os.path.join(util.ROOT, "bench.py")] + tail
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["MASTER_ADDR"] = "127.0.0.1"
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=util.ROOT)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == world and out["steps"] == 2 and out["warmup"] == 1
    assert out["config"]["blocks_per_gpu"] == blocks
    assert "host bounce" in out["config"]["size_gather"] and "LizardGPU_setCollectives" in out["config"]["size_gather"]
    # the line says what carried the exchange and how many ranks it was made for (an RCCL job also says what RCCL itself saw,
    # and bench.py refuses to print a line when that is not --gpus)
    tr = out["config"]["size_gather_transport"]
    assert tr["transport"] == "LizardGPU_setCollectives table" and tr["ranks_requested"] == world and tr["rccl_ranks_seen"] == 0
    assert "scaling=weak" in out["config"]["scaling_note"] and len(out["config"]["workload"]) <= 120
    # BASELINE configs[4] read literally rides in the same line: the SAME 96 blocks of 4 MiB in the whole job, 96 / N per GPU
    strong = [c for c in out["configs"] if c["scaling"] == "strong"]
    assert len(strong) == 1 and strong[0]["blocks_per_gpu"] == 96 // world and strong[0]["block_size"] == 4 << 20
    assert strong[0]["blocks_total"] == world * (96 // world) and strong[0]["blocks_checked"] == strong[0]["blocks_total"]
    assert [p["rank"] for p in strong[0]["per_rank"]] == list(range(world)) and strong[0]["value"] > 0
    assert [p["rank"] for p in out["per_rank"]] == list(range(world))
    assert all(p["kernel_ms"] > 0 and p["gather_us"] > 0 for p in out["per_rank"])
    assert out["cpu_baseline"]["value"] > 0 and out["cpu_baseline"]["cores"] == 1
    assert out["blocks_checked"] == world * blocks and out["blocks_checked_bytes"] == world * blocks
    assert out["value"] > 0 and out["roofline"]["frac"] > 0
    # the whole job's compressed size: every rank's blocks are different blocks (seed = rank * blocks + b)
    want = sum(len(util.oracle_compress(util.datagen(262144, 0.5, 0.0, b), 10)) for b in (0, blocks, world * blocks - 1))
    assert want > 0 and out["compressed_bytes"] > world * blocks * 100000
