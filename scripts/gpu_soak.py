"""One-off GPU soak (run with gpurun from the repo root): thousands of stitched random inputs, every level, one block
per call and in batches, against the oracle.  Not part of the test suite (tests/test_random_parity.py is the bounded form)."""
import os, sys, random, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import util
from test_random_parity import make_case, LEVELS
from lizard_amd import _lib, api
L = _lib.lib()
t0 = time.time(); n = 0; bad = 0
for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 100, (int(sys.argv[1]) if len(sys.argv) > 1 else 100) + 8):
    rng = random.Random(seed)
    for trial in range(500):
        data = make_case(rng, 500000)
        level = rng.choice(LEVELS)
        out, r = util.compress_with(L.Lizard_compress, data, level)
        n += 1
        if out != util.oracle_compress(data, level):
            bad += 1; print('MISMATCH seed', seed, 'trial', trial, 'level', level, 'n', len(data)); 
    # batches
    for trial in range(12):
        level = rng.choice(LEVELS); bs = rng.choice([1000, 4096, 30000, 65536, 131072, 262144, 400000])
        data = b"".join(make_case(rng, 300000) for _ in range(40))
        outs = api.compress_blocks(data, bs, level)
        for i, o in enumerate(outs):
            n += 1
            if o != util.oracle_compress(data[i*bs:(i+1)*bs], level):
                bad += 1; print('BATCH MISMATCH seed', seed, trial, level, bs, i)
    print('seed', seed, 'done', n, 'bad', bad, '%.0fs' % (time.time()-t0), flush=True)
    if time.time() - t0 > (float(sys.argv[2]) if len(sys.argv) > 2 else 200): break
print('TOTAL', n, 'bad', bad)
sys.exit(1 if bad else 0)
