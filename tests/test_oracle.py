"""CPU: pins the oracle restatement (oracle/lizard_oracle.c) against the reference.

(a) golden vectors recorded from the compiled, unmodified reference (tests/golden/reference_vectors.json,
    produced by tests/golden/make_golden.py in the build container);
(b) byte-for-byte against oracle/_ref/liblizard_ref_reset.so whenever that prebuilt library is present;
(c) the SURVEY.md §8c known answers on 64 MiB of datagen P50 (sum of sizes + chained XXH64).
"""
import ctypes
import json
import os

import pytest
import xxhash

import util

with open(os.path.join(util.GOLDEN_DIR, "reference_vectors.json")) as f:
    GOLDEN = json.load(f)

CORPUS = dict(util.corpus())
CORPUS_LONG = dict(util.corpus_long())
LEVELS = [l for l in GOLDEN["levels"] if util.oracle().lzo_level_supported(l)]


def test_levels_in_scope_are_restated():
    for lvl in (10, 11, 12, 13, 14, 15, 16, 17, 20, 21, 22, 30, 31, 32, 33, 34, 35, 36, 37, 38, 40, 41, 42):
        assert lvl in LEVELS


def test_datagen_matches_recorded_inputs():
    for name, data in list(CORPUS.items()) + list(CORPUS_LONG.items()):
        assert util.sha(data) == GOLDEN["cases"][name]["input_sha256"], name


def test_datagen_matches_reference_generator():
    dg = util.reference_datagen()
    if dg is None:
        pytest.skip("oracle/_ref not built here")
    for size, p, seed in [(1 << 20, 0.5, 0), (300001, 0.2, 9), (70000, 0.9, 3), (65536, 1.0, 1), (100, 0.0, 5)]:
        buf = ctypes.create_string_buffer(size)
        dg.RDG_genBuffer(buf, size, p, 0.0, seed)
        assert buf.raw == util.datagen(size, p, 0.0, seed)


@pytest.mark.parametrize("level", LEVELS)
def test_oracle_vs_golden(level):
    for name, data in list(CORPUS.items()) + list(CORPUS_LONG.items()):
        out, r = util.compress_with(util.oracle().lzo_compress, data, level)
        g = GOLDEN["cases"][name]["out"][str(level)]
        assert r == g["size"], (name, level)
        assert util.sha(out) == g["sha256"], (name, level)


@pytest.mark.parametrize("level", [l for l in (10, 21, 30) if l in LEVELS])
def test_oracle_frame_style_capacity(level):
    """maxDstSize = srcSize-1 (reference lib/lizard_frame.c:461): same bytes when it fits, else 0."""
    for name, data in CORPUS.items():
        if len(data) < 2:
            continue
        out, r = util.compress_with(util.oracle().lzo_compress, data, level, cap=len(data) - 1)
        g = GOLDEN["frame_style"][name][str(level)]
        assert r == g["size"], (name, level)
        assert util.sha(out) == g["sha256"], (name, level)


@pytest.mark.parametrize("level", LEVELS)
def test_oracle_vs_compiled_reference(level):
    ref = util.reference()
    if ref is None:
        pytest.skip("oracle/_ref not present")
    for name, data in CORPUS.items():
        a, ra = util.compress_with(util.oracle().lzo_compress, data, level)
        b, rb = util.compress_with(ref.Lizard_compress, data, level)
        assert ra == rb and a == b, (name, level)
    # every possible capacity around the exact size of one block
    data = CORPUS["gen65537_p0.5"]
    exact = len(util.oracle_compress(data, level))
    for cap in (1, 2, 16, exact - 1, exact, exact + 1):
        a, ra = util.compress_with(util.oracle().lzo_compress, data, level, cap=cap)
        b, rb = util.compress_with(ref.Lizard_compress, data, level, cap=cap)
        assert ra == rb and a == b, (level, cap)


@pytest.mark.parametrize("key", sorted(k for k in GOLDEN["p50_64m"] if k.startswith("L")))
def test_oracle_known_answers_p50_64m(key):
    level, bs = int(key[1:].split("_B")[0]), int(key.split("_B")[1])
    if level not in LEVELS:
        pytest.skip("level not restated yet")
    N = 64 << 20
    buf = ctypes.create_string_buffer(N)
    util.oracle().lzo_datagen(buf, N, 0.5, 0.0, 0)
    assert util.sha(buf.raw) == GOLDEN["p50_64m"]["input_sha256"]
    base = ctypes.addressof(buf)
    bound = util.oracle().lzo_compress_bound(bs)
    out = ctypes.create_string_buffer(bound)
    tot, h = 0, 0
    for i in range(N // bs):
        n = util.oracle().lzo_compress(base + i * bs, out, bs, bound, level)
        tot += n
        h = xxhash.xxh64(out.raw[:n], seed=h).intdigest()
    assert tot == GOLDEN["p50_64m"][key]["sum"]
    assert "%016x" % h == GOLDEN["p50_64m"][key]["xxh64_chain"]
