"""TEST INFRASTRUCTURE ONLY: builds tests/emul/libemul.so (the CPU SIMT emulator running the product's
kernel bodies). Never imported by the product package."""
import os, subprocess, sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
OUT = os.path.join(HERE, "libemul.so")
SRCS = [os.path.join(HERE, "simt.cpp"), os.path.join(HERE, "emul_api.cpp")]
DEPS = SRCS + [os.path.join(HERE, "lz_wave.h")] + [
    os.path.join(ROOT, "lizard_amd", "csrc", f) for f in os.listdir(os.path.join(ROOT, "lizard_amd", "csrc")) if f.endswith(".h")]


def build(force=False):
    """LZ_EMUL_DEFS="-DLZ_FAST_128=1": emulate a tuning variant of the kernels (built beside the default library, picked up by
    util.emulator() for the lifetime of that environment variable) — e.g. LZ_EMUL_DEFS=-DLZ_FAST_128=1 python scripts/emul_fuzz.py 1 60 300000 long 10,30"""
    defs = os.environ.get("LZ_EMUL_DEFS", "").split()
    out = OUT if not defs else os.path.join(HERE, "libemul_variant.so")
    stamp = out + ".defs"
    same = not defs or (os.path.exists(stamp) and open(stamp).read() == " ".join(defs))
    def fresh():
        return same and os.path.exists(out) and all(os.path.getmtime(out) >= os.path.getmtime(d) for d in DEPS)
    if not force and fresh():
        return out
    # several test processes may find the library stale at once (pytest-xdist, soaks started together): one of them builds, into a
    # temporary file that is renamed into place (processes that already loaded the old file keep their mapping), the others wait
    import fcntl
    with open(out + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if not force and fresh():
            return out
        tmp = "%s.tmp.%d" % (out, os.getpid())
        cmd = ["g++", "-O2", "-g", "-std=c++17", "-fPIC", "-shared", "-fno-omit-frame-pointer", "-Wall", "-Wextra",
               "-Wno-unused-function", "-Wno-unknown-pragmas", "-pthread", "-I", HERE, "-o", tmp] + defs + SRCS
        subprocess.check_call(cmd)
        os.replace(tmp, out)
        if defs:
            open(stamp, "w").write(" ".join(defs))
    return out


if __name__ == "__main__":
    print(build(force="-f" in sys.argv))
