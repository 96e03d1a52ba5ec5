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
    if not force and os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(d) for d in DEPS):
        return OUT
    cmd = ["g++", "-O2", "-g", "-std=c++17", "-fPIC", "-shared", "-fno-omit-frame-pointer", "-Wall", "-Wextra",
           "-Wno-unused-function", "-Wno-unknown-pragmas", "-pthread", "-I", HERE, "-o", OUT] + SRCS
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force="-f" in sys.argv))
