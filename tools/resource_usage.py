#!/usr/bin/env python3
"""tools/resource_usage.py -- per-kernel register / scratch / occupancy table from the compiler's own remarks.

    python tools/resource_usage.py [--out profiles/rNN_resource_usage.txt] [extra hipcc flags ...]

Compiles curve25519_amd/csrc/engine.hip for gfx950 (device code only, nothing is linked) with
-Rpass-analysis=kernel-resource-usage and prints one line per kernel.  `alloc` is the hardware allocation
(VGPR + AGPR rounded up to the granule of 8), `waves` what the register file allows per SIMD (512 / alloc, at
most 8); the launch bounds and LDS may lower it further (the compiler's own `Occupancy` column has those in).
tests/test_resources.py asserts on the same parse (scratch = 0 for the hot kernels).
"""
import os
import re
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ENGINE = os.path.join(ROOT, "curve25519_amd", "csrc", "engine.hip")
BUILD_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-mllvm", "-pragma-unroll-threshold=131072"]

FIELDS = [("Function Name", "name", str), ("SGPRs", "sgpr", int), ("VGPRs", "vgpr", int), ("AGPRs", "agpr", int),
          ("ScratchSize [bytes/lane]", "scratch", int), ("Occupancy [waves/SIMD]", "occupancy", int),
          ("SGPRs Spill", "sgpr_spill", int), ("VGPRs Spill", "vgpr_spill", int), ("LDS Size [bytes/block]", "lds", int)]


def demangle(names):
    filt = shutil.which("c++filt") or "/opt/rocm/lib/llvm/bin/llvm-cxxfilt"
    try:
        out = subprocess.run([filt], input="\n".join(names), capture_output=True, text=True, check=True).stdout.split("\n")
        return [re.sub(r"\(.*$", "", re.sub(r"^void ", "", o)) for o in out[:len(names)]]
    except Exception:
        return names


def parse_remarks(text):
    """-> list of dicts, one per kernel, in the order the compiler reported them."""
    kernels, cur = [], None
    for line in text.split("\n"):
        m = re.search(r"remark: +(.*?): +(\S+) \[-Rpass-analysis", line)
        if not m:
            continue
        key, val = m.group(1).strip(), m.group(2)
        for label, short, conv in FIELDS:
            if key == label:
                if short == "name":
                    cur = {"name": val}
                    kernels.append(cur)
                elif cur is not None:
                    cur[short] = conv(val)
    for k, d in zip(kernels, demangle([k["name"] for k in kernels])):
        k["pretty"] = d
    return kernels


def compile_remarks(extra_flags=()):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    cmd = [hipcc, *BUILD_FLAGS, *extra_flags, "--cuda-device-only", "-c", ENGINE, "-o", os.devnull,
           "-Rpass-analysis=kernel-resource-usage"]
    p = subprocess.run(cmd, capture_output=True, text=True)
    if p.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + p.stderr[-4000:])
    return parse_remarks(p.stderr)


def table(kernels):
    rows = ["# hipcc -Rpass-analysis=kernel-resource-usage, gfx950, flags: " + " ".join(BUILD_FLAGS),
            "# alloc = ceil((vgpr+agpr)/8)*8; waves_by_regs = min(8, 512 // alloc); occupancy = the compiler's figure",
            f"{'kernel':58s} {'vgpr':>5s} {'agpr':>5s} {'alloc':>5s} {'sgpr':>5s} {'spill_v':>7s} {'scratch_B':>9s} {'lds_B':>7s} {'waves_by_regs':>13s} {'occupancy':>9s}"]
    for k in sorted(kernels, key=lambda k: k["pretty"]):
        alloc = -(-(k.get("vgpr", 0) + k.get("agpr", 0)) // 8) * 8
        waves = min(8, 512 // max(alloc, 8))
        rows.append(f"{k['pretty'][:58]:58s} {k.get('vgpr', 0):5d} {k.get('agpr', 0):5d} {alloc:5d} {k.get('sgpr', 0):5d} "
                    f"{k.get('vgpr_spill', 0):7d} {k.get('scratch', 0):9d} {k.get('lds', 0):7d} {waves:13d} {k.get('occupancy', 0):9d}")
    return "\n".join(rows) + "\n"


def main():
    args = sys.argv[1:]
    out = None
    if "--out" in args:
        i = args.index("--out")
        out = args[i + 1]
        del args[i:i + 2]
    text = table(compile_remarks(args))
    sys.stdout.write(text)
    if out:
        with open(out, "w") as f:
            f.write(text)


if __name__ == "__main__":
    main()
