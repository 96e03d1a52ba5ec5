"""Generates tests/golden/reference_vectors.json from the UNMODIFIED reference library built out of
/root/reference (oracle/_ref/liblizard_ref_reset.so, i.e. -DLIZARD_RESET_MEM: zero-initialised state).

Run in the build container (the reference does not exist on the GPU box):
    python tests/golden/make_golden.py                  (everything; --add-known-answers / --add-levels keep what is recorded and add
                                                         the missing 64 MiB known answers / the levels the cases do not have yet)
Inputs are regenerated deterministically by tests/util.corpus(); the JSON pins sha256(input) too, so a
drift of the generator restatement is detected rather than silently re-blessed.
"""
import json
import os
import sys

import xxhash

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import util  # noqa: E402

LEVELS = [10, 11, 12, 13, 14, 15, 16, 17, 20, 21, 22, 30, 31, 32, 33, 34, 35, 36, 37, 38, 40, 41, 42]
P50_64M_KEYS = [(10, 262144), (11, 262144), (21, 262144), (30, 262144), (10, 4 << 20), (10, 65536), (13, 262144), (15, 262144),
                (17, 262144), (35, 262144), (22, 262144), (31, 262144), (41, 262144), (42, 262144),
                (14, 262144), (16, 262144), (34, 262144), (36, 262144), (37, 262144), (38, 262144),
                (12, 262144), (32, 262144), (33, 262144), (20, 262144), (40, 262144)]
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_vectors.json")


def main():
    ref = util.reference()
    assert ref is not None, "needs /root/reference (build container)"
    dg = util.reference_datagen()
    vec = {"levels": LEVELS, "cases": {}, "frame_style": {}, "p50_64m": {}}
    add_only = "--add-known-answers" in sys.argv          # keep everything recorded, add the missing 64 MiB known answers
    add_levels = "--add-levels" in sys.argv               # keep everything recorded, add the levels of LEVELS the cases do not have yet
    if add_only or add_levels:
        with open(OUT) as f:
            vec = json.load(f)
        vec["levels"] = LEVELS
    for name, data in ([] if add_only else util.corpus() + util.corpus_long()):
        entry = {"n": len(data), "input_sha256": util.sha(data), "out": {}}
        if add_levels and name in vec["cases"]:               # (a case the file does not hold yet is recorded at every level)
            entry = vec["cases"][name]
            assert entry["n"] == len(data) and entry["input_sha256"] == util.sha(data), name
        for lvl in LEVELS:
            if str(lvl) in entry["out"]:
                continue
            out, r = util.compress_with(ref.Lizard_compress, data, lvl)
            entry["out"][str(lvl)] = {"size": r, "sha256": util.sha(out)}
        # frame-style capacity (lizard_frame.c:461: maxDstSize = srcSize-1): 0 when it does not fit
        if 1 < len(data) <= 300000:
            fs = {}
            for lvl in (10, 21, 30):
                out, r = util.compress_with(ref.Lizard_compress, data, lvl, cap=len(data) - 1)
                fs[str(lvl)] = {"size": r, "sha256": util.sha(out)}
            vec["frame_style"][name] = fs
        vec["cases"][name] = entry
    # SURVEY.md §8c known answers: RDG_genBuffer(64 MiB, 0.5, 0.0, seed 0), N x blockSize blocks
    import ctypes
    N = 64 << 20
    buf = ctypes.create_string_buffer(N)
    dg.RDG_genBuffer(buf, N, 0.5, 0.0, 0)
    vec["p50_64m"]["input_sha256"] = util.sha(buf.raw)
    base = ctypes.addressof(buf)
    for lvl, bs in P50_64M_KEYS:
        if f"L{lvl}_B{bs}" in vec["p50_64m"]:
            continue                                   # --add-known-answers: already recorded
        bound = ref.Lizard_compressBound(bs)
        out = ctypes.create_string_buffer(bound)
        tot, h, sizes = 0, 0, []
        for i in range(N // bs):
            n = ref.Lizard_compress(base + i * bs, out, bs, bound, lvl)
            tot += n
            sizes.append(n)
            h = xxhash.xxh64(out.raw[:n], seed=h).intdigest()
        vec["p50_64m"][f"L{lvl}_B{bs}"] = {"sum": tot, "xxh64_chain": "%016x" % h, "first_sizes": sizes[:8]}
        print(lvl, bs, tot, "%016x" % h)
    # frames: LizardF_compressFrame of the zero-state reference build
    cases = dict(util.corpus())
    vec["frames"] = {}
    for name, case, lvl, bsid, crc, csz in util.FRAME_CASES:
        fr = util.reference_frame(cases[case], util.frame_prefs(lvl, bsid, crc, csz))
        vec["frames"][name] = {"size": len(fr), "sha256": util.sha(fr), "head": fr[:16].hex()}
    path = os.path.join(util.GOLDEN_DIR, "reference_vectors.json")
    with open(path, "w") as f:
        json.dump(vec, f, indent=1, sort_keys=True)
    # two tiny raw fixtures for eyeballing byte-level diffs
    small = util.datagen(4096, 0.5, 0.0, 42)
    with open(os.path.join(util.GOLDEN_DIR, "p50_4k_seed42.bin"), "wb") as f:
        f.write(small)
    for lvl in (10, 21, 30):
        out, _ = util.compress_with(ref.Lizard_compress, small, lvl)
        with open(os.path.join(util.GOLDEN_DIR, f"p50_4k_seed42.L{lvl}.liz_block"), "wb") as f:
            f.write(out)
    print("wrote", path)


if __name__ == "__main__":
    main()
