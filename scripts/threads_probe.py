"""Aggregate rate of N host threads calling Lizard_compress (one block per call) at once: python scripts/threads_probe.py [threads] [level] [bytes] [seconds]"""
import ctypes, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import util
from lizard_amd import _lib
L = ctypes.CDLL(_lib.LIB_PATH)
L.Lizard_compress.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
nt = int(sys.argv[1]) if len(sys.argv) > 1 else 64
level = int(sys.argv[2]) if len(sys.argv) > 2 else 10
n = int(sys.argv[3]) if len(sys.argv) > 3 else 262144
secs = float(sys.argv[4]) if len(sys.argv) > 4 else 2.0
bound = util.oracle().lzo_compress_bound(n)
datas = [ctypes.create_string_buffer(util.datagen(n, 0.5, 0.0, t), n) for t in range(nt)]
want = [util.oracle_compress(d.raw, level) for d in datas]
dst0 = ctypes.create_string_buffer(bound)
L.Lizard_compress(datas[0], dst0, n, bound, level)                # context creation outside the timed region
for threads in sorted({1, 2, 8, 16, 32, nt}):
    if threads > nt: continue
    counts = [0] * threads; bad = []
    start = threading.Barrier(threads + 1); stop = [False]
    def worker(t):
        dst = ctypes.create_string_buffer(bound)
        start.wait()
        while not stop[0]:
            r = L.Lizard_compress(datas[t], dst, n, bound, level)
            if r != len(want[t]) or dst.raw[:r] != want[t]: bad.append(t)
            counts[t] += 1
    th = [threading.Thread(target=worker, args=(t,)) for t in range(threads)]
    for x in th: x.start()
    start.wait(); t0 = time.perf_counter(); time.sleep(secs); stop[0] = True
    for x in th: x.join()
    dt = time.perf_counter() - t0
    b, j = ctypes.c_ulonglong(), ctypes.c_ulonglong()
    L.LizardGPU_combinerStats(ctypes.byref(b), ctypes.byref(j))
    print(f"threads {threads:3d} level {level} {n} B: {sum(counts) * n / dt / 1e6:9.1f} MB/s aggregate, {sum(counts)} calls, mismatches {len(bad)}, launches so far {b.value} for {j.value} blocks", flush=True)
