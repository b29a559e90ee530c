#!/usr/bin/env python3
"""tools/isa_mix_report.py engine.s > profiles/rNN_isa_mix.txt -- the hot loops' instruction mix with the MAD share of
their issue cycles (uses tools/isa_mix.py's parser).  engine.s: hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm
-pragma-unroll-threshold=131072 --cuda-device-only -S curve25519_amd/csrc/engine.hip"""
import collections
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import isa_mix  # noqa: E402

text = open(sys.argv[1]).read()
FULL = isa_mix.FULL


def loops(sym):
    name, lines = isa_mix.kernel_body(text, sym)
    labels, insts = {}, []
    for l in lines:
        l = l.split(';')[0].rstrip()
        if not l.strip():
            continue
        m = re.match(r'^(\.LBB\w+):', l)
        if m:
            labels[m.group(1)] = len(insts)
            continue
        if l.startswith('\t') and not l.strip().startswith('.'):
            insts.append(l.strip())
    out = []
    for i, ins in enumerate(insts):
        m = re.match(r's_c?branch\w*\s+(\.LBB\w+)', ins)
        if m and m.group(1) in labels and labels[m.group(1)] <= i:
            out.append((labels[m.group(1)], i))
    return name, insts, out


def mix(insts, lo, hi):
    c = collections.Counter(x.split()[0] for x in insts[lo:hi + 1])
    valu = sum(v for k, v in c.items() if k.startswith('v_'))
    mad = c.get('v_mad_u64_u32', 0)
    full = sum(v for k, v in c.items() if k in FULL or k.replace('_e32', '') in FULL)
    half = valu - mad - full
    cyc = 4 * mad + 4 * half + 2 * full
    return dict(n=hi - lo + 1, valu=valu, mad=mad, half=half, full=full, other=(hi - lo + 1) - valu, share=4 * mad / max(cyc, 1), c=c)


def show(title, sym, pick):
    name, insts, ls = loops(sym)
    print(f"## {title}   [{name}: {len(insts)} instructions]")
    for lo, hi in ls:
        m = mix(insts, lo, hi)
        if not pick(m):
            continue
        top = ", ".join(f"{k} {v}" for k, v in m['c'].most_common(14))
        print(f"loop [{lo}, {hi}]  {m['n']} instructions: VALU {m['valu']} = mad64 {m['mad']} + half-rate {m['half']} + full-rate {m['full']}; "
              f"scalar/memory/other {m['other']};  mad_cycle_share {m['share']:.3f}")
        print(f"    {top}")
    print()


print("# Instruction mix of the hot loops, from hipcc -S of curve25519_amd/csrc/engine.hip (gfx950, the product's flags), tools/isa_mix_report.py.")
print("# Issue classes, nominal: v_mad_u64_u32 4 cycles per wave-instruction per SIMD, the other VOP3 / 64-bit / multiply instructions")
print("# (\"half-rate\") 4, VOP2 adds / ands / subs / moves (\"full-rate\") 2.  Measured in shader cycles at the kernels' four waves per SIMD")
print("# (tools/ubench/mad_peak with s_memtime, profiles/r04_mad_peak.txt): 4.26 for both 4-cycle classes; a VOP2 instruction 2.13 in a")
print("# run of its own kind on all waves and ~4 alone between another wave's MADs; with the runs at low priority (s_setprio) the ladder step")
print("# pays 1.37 per VOP2 instruction on average (profiles/r04_cycle_probe.txt).")
print("# mad_cycle_share = 4*mad / (4*mad + 4*half + 2*full): the most a VALU-bound kernel can reach of the v_mad_u64_u32 roof")
print("# with this instruction stream.\n")
show("X25519 ladder step (5 M + 4 S + a24 + 8 add/sub + select): one trip = one scalar bit", "k_x25519_ladderILb0E", lambda m: 1200 < m['n'] < 1480)   # 1246 VALU + 216 s_setprio + loop control
show("verification walk: digit rounds (the biggest loop is one round: 4 doublings or 4 x (doubling + LDS-row addition), then two table-row "
     "additions from prefetched packed rows; inside it the 3-doubling loop and the 4-step sigma loop)", "k_ed25519_verify_fast_walk", lambda m: m['n'] > 800)
show("verification points kernel: the squaring loops of the square root (99-101 instructions per squaring) and the table-build loop",
     "k_ed25519_verify_fast_points", lambda m: m['n'] > 90)
show("sign: the fixed-base walk's loops (one signed-comb row addition per inner trip)", "k_ed25519_sign_multILb0", lambda m: m['n'] > 500)
