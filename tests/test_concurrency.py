"""GPU: many host threads in the one-block entry points at once (SURVEY.md section 8b "Ownership / threading": the replacement
adds hidden process-wide GPU state and must be safe when many host threads call Lizard_compress concurrently; reference
contract lib/lizard_compress.c:583-606: no globals, one state per thread).

The one-block calls are COMBINED (lizard_amd/csrc/lizard_pipeline_host.c): callers that arrive while a launch is in flight
leave together in the next one.  Checked here: every result is bit-exact against the oracle whatever the mix of levels, sizes
and capacities in a batch; stream create / free churn; LizardGPU_shutdown while callers are in flight; and that combining
really happens (fewer launches than blocks)."""
import ctypes
import random
import threading

import pytest

import util

C = ctypes
LEVELS = [10, 30, 21, 41, 11, 13, 17, 35, 22, 20, 32]
SIZES = [1, 19, 21, 100, 4096, 65537, 131072, 131073, 262144, 300000]


@pytest.fixture(scope="module")
def lib():
    from lizard_amd import _lib
    L = C.CDLL(_lib.LIB_PATH)
    L.Lizard_compress.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
    L.Lizard_compress_extState.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
    L.Lizard_createStream.restype = C.c_void_p; L.Lizard_createStream.argtypes = [C.c_int]
    L.Lizard_freeStream.argtypes = [C.c_void_p]
    L.Lizard_compress_continue.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    L.Lizard_sizeofState.argtypes = [C.c_int]
    L.LizardGPU_combinerStats.argtypes = [C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong)]
    L.LizardGPU_lastError.restype = C.c_char_p
    return L


def _inputs():
    rnd = random.Random(4)
    pool = {}
    for n in SIZES:
        pool[n] = [util.datagen(n, p, 0.0, n * 3 + i) for i, p in enumerate((0.2, 0.5, 0.9))]
    pool[262144].append(rnd.randbytes(262144))                     # incompressible: the raw-fallback rules, and 0 in a tight dst
    return pool


def _stats(L):
    b, j = C.c_ulonglong(), C.c_ulonglong()
    assert L.LizardGPU_combinerStats(C.byref(b), C.byref(j)) == 0
    return b.value, j.value


@pytest.mark.gpu
def test_64_threads_mixed_levels_sizes_capacities(lib):
    pool = _inputs()
    plan = []                                                        # (thread, data, level, cap kind)
    rnd = random.Random(11)
    NT, CALLS = 64, 10
    for t in range(NT):
        for _ in range(CALLS):
            n = rnd.choice(SIZES)
            plan.append((t, rnd.choice(pool[n]), rnd.choice(LEVELS), rnd.choice(("bound", "bound", "frame", "tiny"))))
    want = {}
    for _, data, level, _ in plan:
        key = (id(data), level)
        if key not in want:
            want[key] = util.oracle_compress(data, level)
    errors = []
    b0, j0 = _stats(lib)

    def worker(t):
        state = C.create_string_buffer(lib.Lizard_sizeofState(10) + 16)
        mine = [p for p in plan if p[0] == t]
        for i, (_, data, level, kind) in enumerate(mine):
            full = want[(id(data), level)]
            bound = util.oracle().lzo_compress_bound(len(data))
            cap = {"bound": bound, "frame": max(len(data) - 1, 0), "tiny": max(len(full) - 1, 0)}[kind]
            dst = C.create_string_buffer(bound + 64)
            C.memset(dst, 0x5A, bound + 64)
            if i & 1:
                r = lib.Lizard_compress(data, dst, len(data), cap, level)
            else:
                r = lib.Lizard_compress_extState(C.addressof(state) + (-C.addressof(state)) % 8, data, dst, len(data), cap, level)
            fits = len(full) <= cap or (len(data) == 1 and cap == 0)
            if fits:
                if r != len(full) or dst.raw[:r] != full:
                    errors.append((t, i, level, len(data), kind, r, len(full)))
            elif r != 0:
                errors.append((t, i, level, len(data), kind, r, "expected 0"))
            if dst.raw[bound:bound + 64] != b"\x5a" * 64:              # nothing behind the bound is touched
                errors.append((t, i, "wrote past the bound"))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(NT)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors[:5]
    b1, j1 = _stats(lib)
    nonempty = sum(1 for p in plan if len(p[1]) > 0)
    assert j1 - j0 == nonempty
    assert b1 - b0 < (j1 - j0) * 0.7, "one-block callers were not combined: %d launches for %d blocks" % (b1 - b0, j1 - j0)


@pytest.mark.gpu
def test_stream_churn_and_shutdown_under_load(lib):
    """create / compress_continue / free in every thread, while the main thread shuts the library down again and again: every
    call still returns the oracle's bytes (a shutdown waits for the batch under way and the next call re-creates the context)."""
    data = [util.datagen(n, 0.5, 0.0, n) for n in (70000, 131073, 262144)]
    want = {(i, lv): util.oracle_compress(d, lv) for i, d in enumerate(data) for lv in (10, 21, 30)}
    errors = []
    stop = threading.Event()

    def worker(t):
        rnd = random.Random(t)
        k = 0
        while not stop.is_set() and k < 40:
            lv = rnd.choice((10, 21, 30))
            i = rnd.randrange(len(data))
            st = lib.Lizard_createStream(lv)
            dst = C.create_string_buffer(len(data[i]) + 1024)
            r = lib.Lizard_compress_continue(st, data[i], dst, len(data[i]), len(data[i]) + 1024)
            if r != len(want[(i, lv)]) or dst.raw[:r] != want[(i, lv)]:
                errors.append((t, k, lv, i, r, lib.LizardGPU_lastError()))
            lib.Lizard_freeStream(st)
            k += 1

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(24)]
    for th in threads:
        th.start()
    for _ in range(6):
        lib.LizardGPU_shutdown()
    for th in threads:
        th.join()
    stop.set()
    assert not errors, errors[:5]
    # and the batch entry still works on the re-created context
    from lizard_amd import api
    blocks = api.compress_blocks(data[2] * 4, 262144, 10)
    assert blocks == [want[(2, 10)]] * 4


@pytest.mark.gpu
def test_leaders_with_more_callers_queued_than_a_batch_holds():
    """The leader of a batch is always a member of it (round-4 advisor finding: with more same-level callers queued in front of a
    waking leader than LZ_ONE_MAX_JOBS, the leader's job used to stay behind while its input was still copied to offset 0 of the
    staging).  A library whose batches hold THREE members (lizard_amd/csrc/Makefile `combiner-test`, built by
    __graft_entry__.build()) under 24 callers: every call compared with the oracle by tests/gpu_threads."""
    import os
    import subprocess
    root = util.ROOT
    var = os.path.join(root, "lizard_amd", "variants", "maxjobs3")
    exe = os.path.join(root, "tests", "gpu_threads")
    if not os.path.exists(os.path.join(var, "liblizard_amd.so")):          # built by __graft_entry__.build(); here: one C file + a link
        subprocess.run(["make", "-C", os.path.join(root, "lizard_amd", "csrc"), "combiner-test"], capture_output=True, timeout=600)
    if not (os.path.exists(os.path.join(var, "liblizard_amd.so")) and os.path.exists(exe)):
        pytest.skip("lizard_amd/variants/maxjobs3/liblizard_amd.so or tests/gpu_threads not built (__graft_entry__.build())")
    env = dict(os.environ, LD_LIBRARY_PATH=var + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    for level, size in ((10, 65537), (21, 20000)):
        r = subprocess.run([exe, "24", str(level), str(size), "1.5"], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout + r.stderr
        lines = [l for l in r.stdout.splitlines() if l.startswith("threads")]
        assert lines and all("mismatches 0," in l for l in lines), r.stdout
        # batches of three: at 24 threads a launch never carries more than 3 blocks
        last = lines[-1].split()
        launches, blocks = int(last[last.index("launches") - 1]), int(last[last.index("blocks") - 1])
        assert blocks <= 3 * launches, lines[-1]
