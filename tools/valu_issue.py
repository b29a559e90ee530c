#!/usr/bin/env python3
"""tools/valu_issue.py -- how busy the vector ALUs are in every kernel of a pass, from the committed rocprofv3 counter passes.

    python tools/valu_issue.py classes engine.s > profiles/rNN_isa_classes.json     # static: instruction classes of the hot loops
    python tools/valu_issue.py report [profiles/rNN_pmc.json [profiles/rNN_isa_classes.json]]     # table on stdout

engine.s:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -pragma-unroll-threshold=131072 --cuda-device-only -S curve25519_amd/csrc/engine.hip

Per kernel (compute(), also what bench.py puts into roofline.*.valu.issue and tests/test_bench_contract.py recomputes):
  simd_cycles          GRBM_GUI_ACTIVE / 8 XCDs: shader cycles the dispatch was resident (every SIMD of the chip sees these)
  valu_insts_per_simd  SQ_INSTS_VALU / 1024 SIMDs
  cycles_per_valu_inst simd_cycles / valu_insts_per_simd -- 4.0 = one VALU instruction issued every four cycles on every SIMD
  valu_busy            SQ_ACTIVE_INST_VALU x 4 / (1024 x simd_cycles): north_star's "VALU-busy against chip peak"
                       (SQ_ACTIVE_INST_* count quad-cycles, MI355X_MICROARCH.md; rocprofv3's VALUBusy for gfx94x is this quotient)
  valu_issue_util      (n4 x 4 + n2 x 2) / simd_cycles with n4 + n2 = valu_insts_per_simd split by the instruction classes of the
                       kernel's hot loop (the largest loop body of the kernel's ISA): n2 = VOP2-encoded full-rate instructions
                       (adds, logic, moves, selects: two cycles when paired with another wave's), n4 = v_mad_u64_u32 and every other
                       VOP3 / 64-bit / multiply instruction (four).  1.0 = the SIMDs issue whatever the stream allows.
The class split is static (a loop body's mix), the counts are dynamic (counters): a kernel whose time is not in its largest loop
(sign_mult: a third is straight-line SHA-512) carries that approximation, stated in `classes_of`."""
import collections
import json
import os
import re
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
import isa_mix  # noqa: E402

SIMDS, XCDS = 1024, 8
# kernels whose classes are extracted (symbol substring -> the name rocprofv3 prints, as tools/rocpd_summary.py shortens it)
KERNELS = {
    "k_x25519_ladderILb0E": "k_x25519_ladder<false>",
    "k_ed25519_verify_fast_walk": "k_ed25519_verify_fast_walk",
    "k_ed25519_verify_fast_points": "k_ed25519_verify_fast_points",
    "k_ed25519_verify_fast_scalars": "k_ed25519_verify_fast_scalars",
    "k_ed25519_sign_multILb0ELb1E": "k_ed25519_sign_mult<false, true>",
    "k_ed25519_sign_finish": "k_ed25519_sign_finish",
    "k_batch_invertI10FinishPackLi16E": "k_batch_invert<FinishPack, 16>",
    "k_batch_invertI13FinishX25519Li16E": "k_batch_invert<FinishX25519, 16>",
}


def _mix(insts):
    c = collections.Counter(x.split()[0] for x in insts)
    valu = sum(v for k, v in c.items() if k.startswith("v_"))
    mad = c.get("v_mad_u64_u32", 0) + c.get("v_mad_i64_i32", 0)
    full = sum(v for k, v in c.items() if k in isa_mix.FULL or k.replace("_e32", "") in isa_mix.FULL)
    return {"instructions": len(insts), "valu": valu, "mad64": mad, "other_four_cycle": valu - mad - full, "vop2": full}


def classes(path):
    text = open(path).read()
    out = {}
    for sym, name in KERNELS.items():
        try:
            _, lines = isa_mix.kernel_body(text, sym)
        except SystemExit:
            continue
        labels, insts = {}, []
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
        loops = []
        for i, ins in enumerate(insts):
            m = re.match(r"s_c?branch\w*\s+(\.LBB\w+)", ins)
            if m and m.group(1) in labels and labels[m.group(1)] <= i:
                loops.append((labels[m.group(1)], i))
        rec = {"whole_kernel": _mix(insts)}
        if loops:
            lo, hi = max(loops, key=lambda r: r[1] - r[0])
            rec["hot_loop"] = _mix(insts[lo:hi + 1])
        out[name] = rec
    return out


def compute(pmc_rec, cls_rec):
    """pmc_rec: one kernel's record of profiles/rNN_pmc.json; cls_rec: the same kernel's record of rNN_isa_classes.json (or None)."""
    simd_cycles = pmc_rec["GRBM_GUI_ACTIVE"] / XCDS
    per_simd = pmc_rec["SQ_INSTS_VALU"] / SIMDS
    out = {"simd_cycles": round(simd_cycles), "valu_insts_per_simd": round(per_simd),
           "cycles_per_valu_inst": round(simd_cycles / per_simd, 3) if per_simd else None,
           "valu_busy": round(pmc_rec["SQ_ACTIVE_INST_VALU"] * 4 / (SIMDS * simd_cycles), 4)}
    if cls_rec:
        m = cls_rec.get("hot_loop") or cls_rec["whole_kernel"]
        vop2_share = m["vop2"] / m["valu"]
        out.update({"vop2_share": round(vop2_share, 4), "mad64_share": round(m["mad64"] / m["valu"], 4),
                    "classes_of": "hot loop" if "hot_loop" in cls_rec else "whole kernel",
                    "valu_issue_util": round(per_simd * ((1 - vop2_share) * 4 + vop2_share * 2) / simd_cycles, 4)})
    return out


def latest(pattern):
    import glob
    hits = sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)))
    return hits[-1] if hits else None


def find(d, name):
    """the record of kernel `name` ("k_batch_invert<FinishPack>" matches "void k_batch_invert<FinishPack, 16>")"""
    hits = [v for k, v in d.items() if name.rstrip(">") in k and "<true>" not in k]
    return hits[0] if hits else None


def for_pass(kernels, pmc_path=None, cls_path=None):
    """{kernel: compute(...)} for the kernels of one pass + the pass's figures weighted by the kernels' resident cycles"""
    pmc_path = pmc_path or latest("r[0-9][0-9]_pmc.json")
    cls_path = cls_path or latest("r[0-9][0-9]_isa_classes.json")
    if not pmc_path:
        return None
    pmc = json.load(open(pmc_path))
    cls = json.load(open(cls_path)) if cls_path else {}
    per, tot_c, busy, util, util_c = {}, 0.0, 0.0, 0.0, 0.0
    for k in kernels:
        rec = find(pmc, k)
        if not rec:
            continue
        r = compute(rec, find(cls, k))
        per[k] = r
        tot_c += r["simd_cycles"]
        busy += r["valu_busy"] * r["simd_cycles"]
        if "valu_issue_util" in r:
            util += r["valu_issue_util"] * r["simd_cycles"]
            util_c += r["simd_cycles"]
    if not per:
        return None
    return {"per_kernel": per, "valu_busy": round(busy / tot_c, 4), "valu_issue_util": round(util / util_c, 4) if util_c else None,
            "weighted_by": "resident SIMD cycles of the pass's kernels",
            "source": f"profiles/{os.path.basename(pmc_path)}" + (f" + profiles/{os.path.basename(cls_path)}" if cls_path else "") +
                      " (committed counter passes of this bench, tools/valu_issue.py; not this run)"}


if __name__ == "__main__":
    if len(sys.argv) >= 3 and sys.argv[1] == "classes":
        json.dump(classes(sys.argv[2]), sys.stdout, indent=1)
        print()
    else:
        pmc_path = sys.argv[2] if len(sys.argv) > 2 else latest("r[0-9][0-9]_pmc.json")
        cls_path = sys.argv[3] if len(sys.argv) > 3 else latest("r[0-9][0-9]_isa_classes.json")
        pmc = json.load(open(pmc_path))
        cls = json.load(open(cls_path)) if cls_path else {}
        print(f"# tools/valu_issue.py report {pmc_path} {cls_path}")
        print(f"{'kernel':<44}{'simd_cycles':>13}{'valu/simd':>11}{'cyc/inst':>9}{'valu_busy':>10}{'vop2':>7}{'issue_util':>11}")
        for k in sorted(pmc):
            if not pmc[k].get("SQ_INSTS_VALU") or pmc[k]["GRBM_GUI_ACTIVE"] < 1e5:
                continue
            r = compute(pmc[k], find(cls, k.replace("void ", "")))
            print(f"{k[:43]:<44}{r['simd_cycles']:>13}{r['valu_insts_per_simd']:>11}{r['cycles_per_valu_inst']:>9}{r['valu_busy']:>10}"
                  f"{r.get('vop2_share', ''):>7}{r.get('valu_issue_util', ''):>11}")
