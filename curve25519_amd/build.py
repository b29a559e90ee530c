"""Builds the gfx950 shared library (hipcc cross-compiles without a GPU).

    python -m curve25519_amd.build          # -> curve25519_amd/libcurve25519_amd.so
    python -m curve25519_amd.build --probe  # -> curve25519_amd/libcurve25519_amd_probe.so, the measurement build
                                            #    (-DC25519_CYCLE_PROBE=1: s_memtime stamps in the X25519 kernels, read by
                                            #    tools/cycle_probe.py and bench.py; never loaded by the product)

The .so is git-ignored but travels with gpurun snapshots; it is rebuilt when any source under
csrc/ or include/ is newer.
"""
import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "libcurve25519_amd.so")
PROBE_LIB = os.path.join(PKG, "libcurve25519_amd_probe.so")
ARCH = "gfx950"


def _sources():
    out = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC))]
    out += [os.path.join(ROOT, "include", f) for f in sorted(os.listdir(os.path.join(ROOT, "include")))]
    return out


def is_stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(s) > t for s in _sources())


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not is_stale():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: the gfx950 library cannot be built on this machine")
    tmp = f"{LIB}.tmp.{os.getpid()}"       # several ranks of one node may find the library stale at the same time: each
                                            # builds into a file of its own and the rename is atomic
    # -pragma-unroll-threshold: k_batch_invert keeps 16 elements and their prefix products in registers, which needs its
    # three 16-trip loops fully unrolled; the default threshold refuses the third one and the arrays land in scratch
    cmd = [hipcc, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-shared", "-mllvm", "-pragma-unroll-threshold=131072",
           os.path.join(CSRC, "engine.hip"), os.path.join(CSRC, "multi_device.hip"), "-ldl", "-o", tmp]
    if verbose:
        cmd.append("-Rpass-analysis=kernel-resource-usage")
        print(" ".join(cmd), file=sys.stderr)
    try:
        subprocess.check_call(cmd)
        os.replace(tmp, LIB)
    finally:
        if os.path.exists(tmp):
            os.remove(tmp)
    return LIB


def build_probe(force: bool = False, level: int = 1) -> str:
    """the engine alone with the in-kernel cycle probe compiled in (engine.hip: C25519_CYCLE_PROBE)"""
    if not force and os.path.exists(PROBE_LIB) and all(os.path.getmtime(s) <= os.path.getmtime(PROBE_LIB) for s in _sources()):
        return PROBE_LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: the gfx950 library cannot be built on this machine")
    tmp = f"{PROBE_LIB}.tmp.{os.getpid()}"
    try:
        subprocess.check_call([hipcc, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-shared", "-mllvm",
                               "-pragma-unroll-threshold=131072", f"-DC25519_CYCLE_PROBE={level}",
                               os.path.join(CSRC, "engine.hip"), "-o", tmp])
        os.replace(tmp, PROBE_LIB)
    finally:
        if os.path.exists(tmp):
            os.remove(tmp)
    return PROBE_LIB


if __name__ == "__main__":
    if "--probe" in sys.argv:
        print(build_probe(force="--force" in sys.argv))
    else:
        print(build(force="--force" in sys.argv, verbose=True))
