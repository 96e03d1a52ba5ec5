"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol include/lizard_amd.h
declares. No compute calls (no GPU here). Also: the product must not link or reference oracle/."""
import ctypes
import os
import re
import subprocess

import pytest

import util
from lizard_amd import _lib, api


@pytest.fixture(scope="module")
def lib():
    _lib.build()
    return ctypes.CDLL(_lib.LIB_PATH)


def declared_functions():
    text = open(os.path.join(util.ROOT, "include", "lizard_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(Lizard(?:GPU|F)?_\w+)\s*\(", text)))


def test_exports_every_declared_symbol(lib):
    names = declared_functions()
    assert "Lizard_compress" in names and "LizardGPU_compressBlocks_device" in names and "LizardF_compressUpdate" in names
    assert "Lizard_decompress_safe_usingDict" in names and "Lizard_XXH64_digest" in names
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/lizard_amd.h but not exported"


def exported():
    out = subprocess.check_output(["nm", "-D", "--defined-only", _lib.LIB_PATH]).decode()
    return sorted(line.split()[-1] for line in out.splitlines() if line.strip())


def test_exports_nothing_else(lib):
    """The dynamic symbol table is the declared surface and nothing more (lizard_amd/csrc/exports.map): no internal seam
    (lzk_*, lzgpu_*) leaks into a process that links the library as its liblizard."""
    assert exported() == declared_functions()


def test_exports_cover_the_reference_library(lib):
    """Every name of the reference's own export list (lib/dll/liblizard.def), of lib/lizard_frame.h, and every Lizard_* /
    LizardF_* / Lizard_XXH* symbol the reference's programs import (recorded with nm -u from programs/*.o, tests/fuzzer.o,
    frametest.o, fullbench.o)."""
    need = """Lizard_compress Lizard_compressBound Lizard_compress_continue Lizard_compress_extState Lizard_createStream
      Lizard_createStreamDecode Lizard_decompress_safe Lizard_decompress_safe_continue Lizard_decompress_safe_partial
      Lizard_decompress_safe_usingDict Lizard_freeStream Lizard_freeStreamDecode Lizard_loadDict Lizard_resetStream Lizard_saveDict
      Lizard_setStreamDecode Lizard_sizeofState
      Lizard_compress_MinLevel Lizard_compress_extState_MinLevel Lizard_createStream_MinLevel Lizard_resetStream_MinLevel
      Lizard_sizeofState_MinLevel Lizard_decompress_safe_forceExtDict Lizard_versionNumber
      LizardF_isError LizardF_getErrorName LizardF_compressFrameBound LizardF_compressFrame LizardF_createCompressionContext
      LizardF_freeCompressionContext LizardF_compressBegin LizardF_compressBound LizardF_compressUpdate LizardF_flush LizardF_compressEnd
      LizardF_createDecompressionContext LizardF_freeDecompressionContext LizardF_getFrameInfo LizardF_decompress
      Lizard_XXH32 Lizard_XXH32_reset Lizard_XXH32_update Lizard_XXH32_digest Lizard_XXH64 Lizard_XXH64_reset Lizard_XXH64_update
      Lizard_XXH64_digest""".split()
    have = set(exported())
    assert not [n for n in need if n not in have]
    if os.path.exists("/root/reference/lib/dll/liblizard.def"):
        text = open("/root/reference/lib/dll/liblizard.def").read()
        names = [l.strip() for l in text.split("EXPORTS")[1].splitlines() if l.strip()]
        assert len(names) == 17 and not [n for n in names if n not in have]


def test_bound_and_version_without_gpu(lib):
    lib.Lizard_compressBound.argtypes = [ctypes.c_int]
    for n in (0, 1, 131071, 131072, 262144, 4 << 20, 0x7E000000):
        assert lib.Lizard_compressBound(n) == util.oracle().lzo_compress_bound(n) == api.Lizard_compressBound(n)
    assert lib.Lizard_compressBound(0x7E000001) == 0
    assert lib.Lizard_versionNumber() == 10000
    assert lib.LizardGPU_levelSupported(10) == 1
    assert lib.Lizard_sizeofState(10) > 0


def test_product_does_not_reference_oracle(lib):
    out = subprocess.check_output(["ldd", _lib.LIB_PATH]).decode()
    assert "oracle" not in out
    for f in os.listdir(os.path.join(util.ROOT, "lizard_amd", "csrc")) + os.listdir(os.path.join(util.ROOT, "lizard_amd")):
        p = os.path.join(util.ROOT, "lizard_amd", "csrc", f)
        if not os.path.isfile(p):
            p = os.path.join(util.ROOT, "lizard_amd", f)
        if os.path.isfile(p) and p.endswith((".h", ".hip", ".c", ".py")):
            assert "oracle/" not in open(p).read().replace("oracle/ ", ""), p


def test_no_gpu_means_loud_failure(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    lib.Lizard_compress.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    dst = ctypes.create_string_buffer(1000)
    assert lib.Lizard_compress(b"a" * 100, dst, 100, 1000, 10) == 0   # reference: 0 == failure
    lib.LizardGPU_lastError.restype = ctypes.c_char_p
    assert lib.LizardGPU_lastError() != b""


REF = "/root/reference"


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "programs")), reason="reference checkout not present (GPU box)")
def test_reference_cli_links_against_library(lib, tmp_path):
    """INTEGRATION.md section 1, literally: the reference's own CLI objects (bench, lizardio, lizardcli, datagen) linked
    against liblizard_amd.so and nothing else — block compressor, decoder, frame layer and xxhash are the product's.
    Everything is compiled into a temporary directory from the sources where they lie.
    Without a GPU the library must fail loudly and the frame layer then stores the blocks raw: the file still
    round-trips (through this library's frame decoder)."""
    import torch
    srcs = ["programs/bench.c", "programs/lizardio.c", "programs/lizardcli.c", "programs/datagen.c"]   # and NO object of the reference's lib/
    objs = []
    for f in srcs:
        o = str(tmp_path / (os.path.basename(f)[:-2] + ".o"))
        subprocess.check_call(["gcc", "-O1", "-w", "-I", REF + "/lib", "-I", REF + "/lib/xxhash", "-I", REF + "/programs",
                               "-DXXH_NAMESPACE=Lizard_", "-c", os.path.join(REF, f), "-o", o])
        objs.append(o)
    exe = str(tmp_path / "lizard_on_gpu_lib")
    libdir = os.path.dirname(_lib.LIB_PATH)
    subprocess.check_call(["gcc", "-o", exe] + objs + ["-L", libdir, "-llizard_amd", "-Wl,-rpath," + libdir])
    out = subprocess.run([exe, "--version"], capture_output=True, text=True)
    assert "Lizard command line interface" in out.stdout + out.stderr
    if torch.cuda.is_available():
        return
    data = util.datagen(300000, 0.5, 0.0, 4)
    (tmp_path / "in.bin").write_bytes(data)
    r = subprocess.run([exe, "-10", "-f", str(tmp_path / "in.bin"), str(tmp_path / "out.liz")], capture_output=True, text=True)
    assert r.returncode == 0 and "no CPU fallback" in r.stderr          # loud, and the frame layer stored raw blocks
    r = subprocess.run([exe, "-d", "-f", str(tmp_path / "out.liz"), str(tmp_path / "back.bin")], capture_output=True, text=True)
    assert r.returncode == 0 and (tmp_path / "back.bin").read_bytes() == data


@pytest.mark.gpu
def test_reference_cli_runs_on_the_gpu_library(tmp_path):
    """The reference's own CLI, built against liblizard_amd.so by oracle/Makefile (oracle/_ref/lizard_cli_amd,
    travels to the GPU box as a binary): compress a file with it, decode it with the reference decoder inside the
    same binary, and check that the frame is byte for byte what the oracle composition predicts for the header
    the CLI chose — i.e. every block really went through the GPU path and came out bit-exact."""
    exe = os.path.join(util.ROOT, "oracle", "_ref", "lizard_cli_amd")
    if not os.path.exists(exe):
        util.need_ref("oracle/_ref/lizard_cli_amd")
    data = util.datagen((9 << 20) + 12345, 0.5, 0.0, 17)
    (tmp_path / "in.bin").write_bytes(data)
    for level, extra in ((10, []), (30, ["-B1"]), (21, ["-B2"])):
        liz, back = tmp_path / f"out{level}.liz", tmp_path / f"back{level}.bin"
        r = subprocess.run([exe, f"-{level}", "-f"] + extra + [str(tmp_path / "in.bin"), str(liz)], capture_output=True, text=True)
        assert r.returncode == 0 and "no CPU fallback" not in r.stderr, r.stderr
        frame = liz.read_bytes()
        assert len(frame) < 0.75 * len(data)                         # compressed, not stored raw
        flg, bd = frame[4], frame[5]
        assert (flg >> 5) & 1 == 1                                   # independent blocks (CLI default)
        want = util.compose_frame(data, level, (bd >> 4) & 7, (flg >> 2) & 1, (flg >> 3) & 1, util.oracle_compress)
        assert frame == want, level
        r = subprocess.run([exe, "-d", "-f", str(liz), str(back)], capture_output=True, text=True)
        assert r.returncode == 0 and back.read_bytes() == data


@pytest.mark.gpu
def test_reference_cli_one_byte_tail_block(tmp_path):
    """A file of k * blockSize + 1 bytes ends in a 1-byte block.  The reference's room test wraps there (maxDstSize 0,
    lib/lizard_compress.c:238) and the frame layer gets a 6-byte "compressed" block instead of a raw one; the drop-in
    path must write the same .liz as the stock library (and as LizardGPU_compressFrame)."""
    exe = os.path.join(util.ROOT, "oracle", "_ref", "lizard_cli_amd")
    if not os.path.exists(exe):
        util.need_ref("oracle/_ref/lizard_cli_amd")
    for k, extra in ((2, ["-B1"]), (1, ["-B2"]), (0, [])):
        bs = {"-B1": 128 << 10, "-B2": 256 << 10}.get(extra[0] if extra else "", 4 << 20)
        data = util.datagen(k * bs + 1, 0.5, 0.0, 23 + k)
        (tmp_path / "in.bin").write_bytes(data)
        liz, back = tmp_path / "out.liz", tmp_path / "back.bin"
        r = subprocess.run([exe, "-10", "-f"] + extra + [str(tmp_path / "in.bin"), str(liz)], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        frame = liz.read_bytes()
        flg, bd = frame[4], frame[5]
        want = util.compose_frame(data, 10, (bd >> 4) & 7, (flg >> 2) & 1, (flg >> 3) & 1, util.oracle_compress)
        assert frame == want, k
        assert frame[-4 - 4 * ((flg >> 2) & 1) - 10:][:4] == bytes([6, 0, 0, 0])        # last block record: LE32 6, not raw
        r = subprocess.run([exe, "-d", "-f", str(liz), str(back)], capture_output=True, text=True)
        assert r.returncode == 0 and back.read_bytes() == data
        # the library's own frame entry agrees
        from lizard_amd import _lib
        L = ctypes.CDLL(_lib.LIB_PATH)
        L.LizardGPU_compressFrameBound.restype = ctypes.c_size_t
        L.LizardGPU_compressFrameBound.argtypes = [ctypes.c_size_t, ctypes.c_void_p]
        L.LizardGPU_compressFrame.restype = ctypes.c_size_t
        L.LizardGPU_compressFrame.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_void_p]
        p = util.frame_prefs(10, (bd >> 4) & 7, (flg >> 2) & 1, (flg >> 3) & 1)
        cap = L.LizardGPU_compressFrameBound(len(data), ctypes.byref(p))
        dst = ctypes.create_string_buffer(cap)
        n = L.LizardGPU_compressFrame(dst, cap, data, len(data), ctypes.byref(p))
        assert dst.raw[:n] == frame, k


def test_reference_frametest_without_a_device_exercises_the_frame_state_machines():
    """The reference's tests/frametest.c linked against liblizard_amd.so alone also runs where there is no GPU: LizardF_compress*
    then stores every block raw (the reference's own behaviour on a 0 from the block compressor, lib/lizard_frame.c:463-467), and
    everything else the program checks is host code of this library — frame headers, every preference, random segmentation of
    Update / flush / End, the bound contracts, LizardF_decompress under random input and output buffer sizes, skippable frames,
    content checksums, error reporting.  30 000 of its randomised tests, two seeds."""
    exe = os.path.join(util.ROOT, "oracle", "_ref", "frametest_amd")
    if not os.path.exists(exe):
        util.reference()                                                     # builds oracle/_ref when the checkout is present
    if not os.path.exists(exe):
        util.need_ref("oracle/_ref/frametest_amd")
    from lizard_amd import _lib
    _lib.build()
    for seed in (4242, 7254):
        r = subprocess.run([exe, "-s%d" % seed, "-i15000"], capture_output=True, text=True, timeout=600)
        tail = (r.stdout + r.stderr).replace("\r", "\n")[-1500:]
        assert r.returncode == 0 and "All tests completed" in tail, tail


@pytest.mark.gpu
@pytest.mark.parametrize("prog,args", [("fuzzer_amd", ["-T10s"]), ("frametest_amd", ["-T10s"]), ("fullbench_amd", ["-i1"])])
def test_reference_test_programs_on_the_gpu_library(prog, args, tmp_path):
    """SURVEY.md section 7 step 2 acceptance: the reference's own fuzzer (tests/fuzzer.c: bounds behaviour, limited
    output, dictionaries, streaming; levels 10 and 17), frametest (tests/frametest.c: every frame preference,
    linked and independent blocks, random segmentation; levels 10-49 clamp/dispatch) and fullbench (tests/fullbench.c:
    every public compression and decompression entry point on a file) built by oracle/Makefile against liblizard_amd.so
    ALONE — no object of the reference's lib/ is linked.  They check round trips and error behaviour, not bytes; linked
    mode is served by history-free blocks (include/lizard_amd.h, Lizard_compress_continue)."""
    exe = os.path.join(util.ROOT, "oracle", "_ref", prog)
    if not os.path.exists(exe):
        util.need_ref("oracle/_ref/%s" % prog)
    ldd = subprocess.check_output(["ldd", exe]).decode()
    assert "liblizard_amd.so" in ldd and "liblizard_ref" not in ldd
    undefined = subprocess.check_output(["nm", "-u", exe]).decode()
    assert {"fuzzer_amd": "Lizard_decompress_safe_continue", "frametest_amd": "LizardF_decompress",
            "fullbench_amd": "Lizard_decompress_safe_forceExtDict"}[prog] in undefined        # decoder and frames come from the product
    if prog == "fullbench_amd":
        f = tmp_path / "in.bin"
        f.write_bytes(util.datagen(3 << 20, 0.5, 0.0, 3))
        args = args + [str(f)]
    r = subprocess.run([exe] + args, capture_output=True, text=True, timeout=900)
    tail = (r.stdout + r.stderr)[-2000:]
    assert r.returncode == 0, tail
    assert "no CPU fallback" not in r.stderr or prog == "frametest_amd", tail


@pytest.mark.gpu
def test_liz_files_equal_the_stock_cli(tmp_path):
    """The reference's CLI linked against the product alone writes, at levels 10 / 21 / 30 and several block sizes, the
    very .liz file the STOCK CLI (reference library inside, oracle/_ref/lizard_cli_ref) writes — and each decodes the other's."""
    amd = os.path.join(util.ROOT, "oracle", "_ref", "lizard_cli_amd")
    ref = os.path.join(util.ROOT, "oracle", "_ref", "lizard_cli_ref")
    if not (os.path.exists(amd) and os.path.exists(ref)):
        util.need_ref("oracle/_ref CLIs")
    data = util.datagen((5 << 20) + 4321, 0.5, 0.0, 41)
    (tmp_path / "in.bin").write_bytes(data)
    for level, extra in ((10, []), (21, ["-B2"]), (30, ["-B1"]), (10, ["-B3", "--content-size"]), (30, ["--no-frame-crc"])):
        a, b = tmp_path / "a.liz", tmp_path / "b.liz"
        for exe, out in ((amd, a), (ref, b)):
            r = subprocess.run([exe, f"-{level}", "-f"] + extra + [str(tmp_path / "in.bin"), str(out)], capture_output=True, text=True)
            assert r.returncode == 0, r.stderr
        assert a.read_bytes() == b.read_bytes(), (level, extra)
        for exe, src in ((amd, b), (ref, a)):
            back = tmp_path / "back.bin"
            r = subprocess.run([exe, "-d", "-f", str(src), str(back)], capture_output=True, text=True)
            assert r.returncode == 0 and back.read_bytes() == data, (level, extra, r.stderr)
    # linked blocks (-BD): valid, decodable by the stock decoder; the bytes are those of independent blocks
    a = tmp_path / "linked.liz"
    r = subprocess.run([amd, "-10", "-BD", "-B1", "-f", str(tmp_path / "in.bin"), str(a)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert (a.read_bytes()[4] >> 5) & 1 == 0
    back = tmp_path / "back.bin"
    r = subprocess.run([ref, "-d", "-f", str(a), str(back)], capture_output=True, text=True)
    assert r.returncode == 0 and back.read_bytes() == data
