"""Builds tests/host_emul/libc25519_emul.so: the device headers of curve25519_amd/csrc compiled by g++ against the C
model of the gfx950 primitives (valu_model.h).  TEST INFRASTRUCTURE -- see valu_model.h."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "curve25519_amd", "csrc")
LIB = os.path.join(HERE, "libc25519_emul.so")
HEADERS = ["valu_gfx950.cuh", "safegcd25519.cuh", "fe25519.cuh", "sc25519.cuh", "sha512.cuh", "ge25519.cuh", "x25519.cuh", "lanes.cuh",
           "curve_constants.cuh", "verify_fast.cuh", "coop25519.cuh", "coop_ops.cuh", "quad25519.cuh"]


def build(force: bool = False) -> str:
    srcs = [os.path.join(HERE, f) for f in ("emul.cpp", "valu_model.h", "coop_wave.h")] + [os.path.join(CSRC, h) for h in HEADERS]
    if not force and os.path.exists(LIB) and all(os.path.getmtime(s) <= os.path.getmtime(LIB) for s in srcs):
        return LIB
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unknown-pragmas", "-Wno-unused-function",
           "-include", os.path.join(HERE, "valu_model.h"), "-I", CSRC, "-I", HERE, os.path.join(HERE, "emul.cpp"),
           "-o", LIB + ".tmp", "-lpthread"]
    subprocess.check_call(cmd)
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    print(build(force=True))
