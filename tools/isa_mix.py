#!/usr/bin/env python3
"""tools/isa_mix.py -- instruction mix of one kernel's loops from hipcc's -save-temps assembly.

    python tools/isa_mix.py <file.s> <kernel-symbol-substring> [--loops]

Prints, for every basic-block range that ends in a backward branch (a loop body), the count of each
opcode, and the totals weighted by the issue classes measured by tools/ubench/valu_rates (full-rate VOP2,
half-rate VOP3-class, v_mad_u64_u32).  Used to find non-MAD overhead in the ladder / Edwards walks.
"""
import collections
import re
import sys

FULL = {"v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_mov_b32", "v_cndmask_b32",
        "v_fma_f32", "v_not_b32", "v_accvgpr_read_b32", "v_accvgpr_write_b32"}


def kernel_body(text, sym):
    # up to the function's end marker (a kernel may hold several s_endpgm: early exits)
    m = re.search(r"^(\S*%s\S*):.*?\n(.*?)\n\.Lfunc_end\d+:" % re.escape(sym), text, re.S | re.M)
    if not m:
        raise SystemExit(f"kernel matching {sym!r} not found")
    return m.group(1), m.group(2).split("\n")


def main():
    text = open(sys.argv[1]).read()
    name, lines = kernel_body(text, sys.argv[2])
    labels = {}
    insts = []
    for l in lines:
        l = l.split(";")[0].rstrip()
        if not l.strip():
            continue
        m = re.match(r"^(\.LBB\w+):", l)
        if m:
            labels[m.group(1)] = len(insts)
            continue
        if l.startswith("\t") and not l.strip().startswith("."):
            insts.append(l.strip())
    print(f"{name}: {len(insts)} instructions")
    loops = []
    for i, ins in enumerate(insts):
        m = re.match(r"s_c?branch\w*\s+(\.LBB\w+)", ins)     # big loops close with s_cbranch out + s_branch back
        if m and m.group(1) in labels and labels[m.group(1)] <= i:
            loops.append((labels[m.group(1)], i))
    for lo, hi in loops:
        c = collections.Counter(x.split()[0] for x in insts[lo:hi + 1])
        n = hi - lo + 1
        valu = sum(v for k, v in c.items() if k.startswith("v_"))
        mad = c.get("v_mad_u64_u32", 0)
        full = sum(v for k, v in c.items() if k in FULL or k.replace("_e32", "") in FULL)
        half = valu - mad - full
        print(f"\nloop [{lo}, {hi}] {n} instructions: VALU {valu} = mad64 {mad} + half-rate {half} + full-rate {full}; "
              f"scalar/other {n - valu}")
        for k, v in c.most_common(40):
            print(f"   {k:28s} {v}")


if __name__ == "__main__":
    main()
