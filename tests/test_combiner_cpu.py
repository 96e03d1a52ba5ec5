"""CPU: the one-block combiner of lizard_amd/csrc/lizard_pipeline_host.c (callers that leave together in ragged batches, two batches
in flight, leaders, stragglers' window, quiesce / resume) without a GPU: tests/combiner_fake.c compiles the file together with a fake
HIP runtime and a fake launcher (the oracle compresses the batch, then a sleep like a one-wave kernel) and lets 24 threads of mixed
sizes, levels and capacities run against it while another thread quiesces / frees / resumes the combiner.  Every result is compared
with the oracle; a watchdog turns a hang into a failure.  Also with three-member batches (the round-4 advisor finding: a leader with
more callers queued in front of it than a batch holds) and under ThreadSanitizer."""
import os
import subprocess

import pytest

import util


def _build(tmp_path, name, extra):
    util.oracle()
    exe = str(tmp_path / name)
    cmd = ["gcc", "-O1", "-g", "-std=gnu99", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I" + os.path.join(util.ROOT, "include"),
           "-I" + util.ORACLE_DIR] + extra + [os.path.join(util.ROOT, "tests", "combiner_fake.c"), "-o", exe, "-L" + util.ORACLE_DIR,
           "-llizard_oracle", "-lpthread", "-Wl,-rpath," + util.ORACLE_DIR]
    subprocess.check_call(cmd)
    return exe


@pytest.mark.parametrize("extra,threads", [([], 24), (["-DLZ_ONE_MAX_JOBS=3"], 24), ([], 64)])
def test_combiner_on_a_fake_device(tmp_path, extra, threads):
    exe = _build(tmp_path, "combiner_fake", extra)
    r = subprocess.run([exe, str(threads), "2", "1"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and " 0 mismatches" in r.stdout, r.stdout + r.stderr
    calls = int(r.stdout.split(" calls")[0].split()[-1])
    assert calls > 200, r.stdout


def test_combiner_under_thread_sanitizer(tmp_path):
    try:
        exe = _build(tmp_path, "combiner_fake_tsan", ["-fsanitize=thread"])
    except subprocess.CalledProcessError:
        pytest.skip("no ThreadSanitizer runtime")
    r = subprocess.run([exe, "16", "2", "1"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and " 0 mismatches" in r.stdout and "ThreadSanitizer" not in r.stderr, (r.stdout + r.stderr)[-3000:]
