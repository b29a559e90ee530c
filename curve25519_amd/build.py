"""Builds the gfx950 shared library (hipcc cross-compiles without a GPU).

    python -m curve25519_amd.build          # -> curve25519_amd/libcurve25519_amd.so
    python -m curve25519_amd.build --probe  # -> curve25519_amd/libcurve25519_amd_probe.so, the measurement build
                                            #    (-DC25519_CYCLE_PROBE=1: s_memtime stamps in the X25519 kernels, read by
                                            #    tools/cycle_probe.py and bench.py; never loaded by the product)

The .so is git-ignored but travels with gpurun snapshots; it is rebuilt when any source under
csrc/ or include/ is newer.  The engine's four translation units (csrc/engine_*.hip) and csrc/multi_device.hip are compiled
side by side into curve25519_amd/_obj/ and linked.
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


ENGINE_UNITS = ("engine_x25519", "engine_fixed_base", "engine_verify", "engine_api")     # csrc/engine_common.cuh says which is which
OBJ = os.path.join(PKG, "_obj")             # git-ignored, does not travel: the GPU box gets the linked libraries
FLAGS = [f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-mllvm", "-pragma-unroll-threshold=131072"]
# -pragma-unroll-threshold: k_batch_invert keeps 16 elements and their prefix products in registers, which needs its
# three 16-trip loops fully unrolled; the default threshold refuses the third one and the arrays land in scratch


def _hipcc() -> str:
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: the gfx950 library cannot be built on this machine")
    return hipcc


def _compile(units, force=False, verbose=False):
    """units: [(source stem, object name, extra flags)]; the stale ones are compiled side by side (the engine's four translation
    units take 9-40 s each: ~40 s for all of them instead of 100 s as one), each into a file of its own that is renamed when
    complete (several ranks of one node may find the library stale at the same time)."""
    os.makedirs(OBJ, exist_ok=True)
    newest = max(os.path.getmtime(s) for s in _sources())
    jobs = []
    for stem, obj, extra in units:
        out = os.path.join(OBJ, obj + ".o")
        if not force and os.path.exists(out) and os.path.getmtime(out) >= newest:
            continue
        tmp = f"{out}.tmp.{os.getpid()}"
        cmd = [_hipcc(), *FLAGS, *extra, "-c", os.path.join(CSRC, stem + ".hip"), "-o", tmp]
        if verbose:
            cmd.append("-Rpass-analysis=kernel-resource-usage")
            print(" ".join(cmd), file=sys.stderr)
        jobs.append((subprocess.Popen(cmd), tmp, out, stem))
    failed = []
    for proc, tmp, out, stem in jobs:
        if proc.wait() == 0:
            os.replace(tmp, out)
        else:
            failed.append(stem)
            if os.path.exists(tmp):
                os.remove(tmp)
    if failed:
        raise RuntimeError("hipcc failed for " + ", ".join(failed))
    return [os.path.join(OBJ, obj + ".o") for _, obj, _ in units]


def _link(objs, lib):
    tmp = f"{lib}.tmp.{os.getpid()}"
    try:
        subprocess.check_call([_hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", *objs, "-ldl", "-o", tmp])
        os.replace(tmp, lib)
    finally:
        if os.path.exists(tmp):
            os.remove(tmp)
    return lib


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not is_stale():
        return LIB
    objs = _compile([(u, u, []) for u in ENGINE_UNITS] + [("multi_device", "multi_device", [])], force, verbose)
    return _link(objs, LIB)


def build_probe(force: bool = False, level: int = 1) -> str:
    """the engine alone with the in-kernel cycle probe compiled in (engine_x25519.hip: C25519_CYCLE_PROBE; the other three
    translation units are the product's objects)"""
    if not force and os.path.exists(PROBE_LIB) and all(os.path.getmtime(s) <= os.path.getmtime(PROBE_LIB) for s in _sources()):
        return PROBE_LIB
    objs = _compile([("engine_x25519", f"engine_x25519_probe{level}", [f"-DC25519_CYCLE_PROBE={level}"])]
                    + [(u, u, []) for u in ENGINE_UNITS if u != "engine_x25519"], force)
    return _link(objs, PROBE_LIB)


if __name__ == "__main__":
    if "--probe" in sys.argv:
        print(build_probe(force="--force" in sys.argv))
    else:
        print(build(force="--force" in sys.argv, verbose=True))
