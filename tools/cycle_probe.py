#!/usr/bin/env python3
"""tools/cycle_probe.py -- in-kernel s_memtime measurement of the X25519 kernels on an UN-PROFILED run.

    python tools/cycle_probe.py LIB_PROBE.so [--fused] [--sections LIB_PROBE2.so ENGINE_PROBE2.s] [--n 1048576] [--dump x.npz]
    python tools/cycle_probe.py x.npz [--fused]                      # analyse an earlier dump again, no GPU needed

LIB_PROBE.so is the engine built with -DC25519_CYCLE_PROBE=1 (`python -m curve25519_amd.build --probe` writes
curve25519_amd/libcurve25519_amd_probe.so; =2 for --sections): every wave stamps s_memtime (one tick = one shader cycle,
MI355X_MICROARCH.md) at its phase boundaries, s_memrealtime (100 MHz) at entry and exit, and the hardware slot it ran
on; level 2 also accumulates the ten sections of a ladder step.  By default the shipped two-launch shape is probed
(k_x25519_ladder + k_batch_invert), with --fused the one-launch kernel k_x25519_fused.  Printed:
  * the shader clock the waves ran at = s_memtime ticks / s_memrealtime ticks x 100 MHz (no host timing involved) and,
    as a cross-check, kernel span / HIP-event time;
  * SIMD cycles per ladder step per wave (the cadence at which a SIMD completes ladders / the steps of a ladder)
    against the issue model of the instruction stream (tools/isa_mix.py classes, nominal and measured cycle costs);
  * where the waves' time goes (fused: ladder, barrier wait, the shared inversion, second barrier wait, finish);
  * with --sections: measured share of each section of the step against the model of the instructions between its marks.
bench.py imports measure() / summary() for roofline.valu.issue_model_frac.
"""
import argparse
import collections
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import isa_mix  # noqa: E402

STEPS = 251 + 4 * 0.55            # 251 ladder steps + the opening doubling and the three closing ones (a doubling = 0.55 step)
# the ladder step's instruction stream (profiles/r04_isa_mix.txt): v_mad_u64_u32, other 4-cycle-class VALU, VOP2
STEP_MAD, STEP_HALF, STEP_FULL = 739, 188, 319
# class costs in SIMD cycles per wave-instruction: nominal, and as measured in place by tools/ubench/mad_peak at four
# resident waves (profiles/r04_mad_peak.txt): MAD and the other 4-cycle-class instructions 4.26 -- 4.0 of execution and
# a 0.26 issue bubble --; a VOP2 instruction 2.13 when another wave's VOP2 shares its slot, ~4.2 alone between MADs.
# "floor": the VOP2 instructions paired AND executed inside the 4-cycle class' bubbles as far as those reach -- what the
# stream costs when VOP2 runs wait at low wave priority for each other (valu_gfx950.cuh: C25519_VOP2_RUN_*).
MODELS = {"nominal": (4.0, 2.0), "measured_vop2_paired": (4.26, 2.13), "measured_vop2_unpaired": (4.26, 4.26)}
BUBBLE = 0.26


def model_cycles(name):
    if name == "floor":
        c4, c2 = MODELS["measured_vop2_paired"]
        n4 = STEP_MAD + STEP_HALF
        return n4 * c4 + max(0.0, STEP_FULL * c2 - n4 * BUBBLE)
    c4, c2 = MODELS[name]
    return (STEP_MAD + STEP_HALF) * c4 + STEP_FULL * c2
WORDS = 20


def measure(lib_path, n=1 << 20, fused=False, reps=6, verbose=False):
    """(best HIP-event ms of one call, its raw stamps [waves, WORDS] uint64) of curve25519_dh_CreateSharedKey_dev"""
    import torch
    from curve25519_amd import synth
    vp, sz = C.c_void_p, C.c_size_t
    L = C.CDLL(os.path.abspath(lib_path))
    L.curve25519_dh_CreateSharedKey_dev.argtypes = [vp, vp, vp, sz, vp]
    L.c25519_amd_probe_set.argtypes = [vp]
    assert L.c25519_amd_probe_words() == WORDS
    dev = torch.device("cuda", torch.cuda.current_device())
    sk_np, pk_np = synth.x25519_inputs(n)
    sk, pk = torch.from_numpy(sk_np).to(dev), torch.from_numpy(pk_np).to(dev)
    out = torch.empty((n, 32), dtype=torch.uint8, device=dev)
    buf = torch.zeros(((n + 63) // 64, WORDS), dtype=torch.int64, device=dev)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    L.c25519_amd_tunable_set.argtypes = [C.c_char_p, C.c_long]
    assert L.c25519_amd_tunable_set(b"XF_SPLIT", 0 if fused else 1) == 0          # the probe library's own knob table
    assert L.c25519_amd_tunable_set(b"QUAD_MAX", 0) == 0                           # the stamps are the one-lane kernels': no quads at 2^12 .. 2^15
    try:
        assert L.c25519_amd_probe_set(p(buf)) == 0
        best = None
        for r in range(reps):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            assert L.curve25519_dh_CreateSharedKey_dev(p(out), p(pk), p(sk), n, st) == 0     # keeps the clocks up
            a.record()
            assert L.curve25519_dh_CreateSharedKey_dev(p(out), p(pk), p(sk), n, st) == 0
            b.record()
            torch.cuda.synchronize()
            ms = a.elapsed_time(b)
            if best is None or ms < best[0]:
                best = (ms, buf.cpu().numpy().astype(np.uint64))
            if verbose:
                print(f"  run {r}: {ms:.3f} ms")
        L.c25519_amd_probe_set(None)
    finally:
        L.c25519_amd_tunable_set(b"XF_SPLIT", -1)
        L.c25519_amd_tunable_set(b"QUAD_MAX", -1)
    return best


def decode_hw(x):
    hw, xcc = x & np.uint64(0xffffffff), (x >> np.uint64(32)) & np.uint64(0xf)
    f = lambda lo, w: ((hw >> np.uint64(lo)) & np.uint64((1 << w) - 1)).astype(np.int64)  # noqa: E731
    return dict(slot=f(0, 4), simd=f(4, 2), cu=f(8, 4), sh=f(12, 1), se=f(13, 3), xcc=xcc.astype(np.int64))


def keys(rec):
    hw = decode_hw(rec[:, 6])
    cu_key = ((hw["xcc"] * 8 + hw["se"]) * 2 + hw["sh"]) * 16 + hw["cu"]
    return hw, cu_key, cu_key * 4 + hw["simd"]


def summary(ms, rec):
    """the numbers bench.py reports: clock, SIMD cycles per ladder step, issue-model fractions"""
    t = rec[:, :6].astype(np.int64)
    hw, cu_key, simd_key = keys(rec)
    # s_memtime counters of different shader engines are not synchronised with each other: spans are taken per CU
    spans = np.array([int(t[cu_key == k, 5].max() - t[cu_key == k, 0].min()) for k in np.unique(cu_key)])
    clock = None
    if rec.shape[1] > 18:
        cyc = (t[:, 5] - t[:, 0]).astype(np.float64)
        rt = (rec[:, 18].astype(np.int64) - rec[:, 17].astype(np.int64)).astype(np.float64)
        if rt.max() > 0:
            long_enough = rt > 0.25 * rt.max()
            clock = float(np.median(cyc[long_enough] / rt[long_enough]) * 0.1)
    # A SIMD serves its resident waves oldest first: ladders complete one after the other at a fixed cadence, whatever
    # the number of resident waves.  The cadence (time between consecutive ladder completions on one SIMD, the first
    # four left out) is the SIMD time one ladder costs.
    cad = np.concatenate([np.diff(np.sort(t[simd_key == k, 1]))[4:] for k in np.unique(simd_key)])
    # SIMD time per ladder: a CU's span over the waves each of its SIMDs served (the cadence says the same while the SIMD
    # serves its waves strictly one after the other; with priority changes inside the step it no longer does)
    per_simd = np.bincount(np.unique(simd_key, return_inverse=True)[1])
    per_step = float(np.median(spans)) / float(np.median(per_simd)) / STEPS if len(per_simd) and np.median(per_simd) >= 8 else None
    out = {"kernel_span_Mcycles": round(float(np.median(spans)) / 1e6, 3), "event_ms": round(ms, 4),
           "shader_clock_GHz": round(clock, 3) if clock else None,
           "shader_clock_GHz_span_over_event_time": round(float(np.median(spans)) / (ms * 1e6), 3),
           "simd_cycles_per_ladder_step": round(per_step, 1) if per_step else None,
           "ladder_step_instructions": {"v_mad_u64_u32": STEP_MAD, "other_4_cycle_class": STEP_HALF, "vop2": STEP_FULL}}
    if per_step:
        for name in list(MODELS) + ["floor"]:
            out["issue_model_frac_" + name] = round(model_cycles(name) / per_step, 4)
        out["issue_model_cycles_floor"] = round(model_cycles("floor"), 1)
        out["vop2_cycles_implied"] = round((per_step - (STEP_MAD + STEP_HALF) * MODELS["measured_vop2_paired"][0]) / STEP_FULL, 3)
        out["ladder_cadence_Mcycles"] = round(float(np.median(cad)) / 1e6, 4) if len(cad) else None
    return out, cad, spans


def report_phases(ms, rec, fused):
    t = rec[:, :6].astype(np.int64)
    hw, cu_key, simd_key = keys(rec)
    s, cad, spans = summary(ms, rec)
    name = "k_x25519_fused" if fused else "k_x25519_ladder (+ k_batch_invert behind it)"
    print(f"\n== {name}, n = {len(t) * 64}, un-profiled: {ms:.3f} ms per call (HIP events), {len(t)} waves on {len(set(cu_key))} CUs / {len(set(simd_key))} SIMDs")
    print(f"kernel span per CU in shader cycles: min {spans.min() / 1e6:.3f} M, median {np.median(spans) / 1e6:.3f} M, max {spans.max() / 1e6:.3f} M")
    print(f"shader clock of the un-profiled launch: {s['shader_clock_GHz']} GHz (s_memtime / s_memrealtime per wave, median); "
          f"span / HIP-event time {s['shader_clock_GHz_span_over_event_time']} GHz" + ("" if fused else " (the event time includes k_batch_invert: a lower bound)"))
    ladder, wait1, inv, wait2, fin = t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2], t[:, 4] - t[:, 3], t[:, 5] - t[:, 4]
    print(f"ladder phase of a wave (wall): median {np.median(ladder) / 1e6:.3f} M cycles; ladder completions on one SIMD are {np.median(cad) / 1e6:.4f} M cycles apart"
          f" (p10 {np.percentile(cad, 10) / 1e6:.4f}, p90 {np.percentile(cad, 90) / 1e6:.4f})")
    print(f"SIMD cycles per ladder step per wave: {s['simd_cycles_per_ladder_step']:.0f} (CU span / waves per SIMD / steps)" + ("   (fused: includes the inversion's share of the SIMD)" if fused else ""))
    for mname, (cm, cf) in MODELS.items():
        print(f"   issue model [{mname}: MAD / 4-cycle class {cm:.2f}, VOP2 {cf:.2f} cycles]: {model_cycles(mname):.0f} -> issue_model_frac {s['issue_model_frac_' + mname]:.3f}")
    print(f"   issue model [floor: 4.26 per 4-cycle-class instruction, VOP2 paired (2.13) and inside the 0.26-cycle bubbles of the former as far as they reach]: "
          f"{model_cycles('floor'):.0f} -> issue_model_frac {s['issue_model_frac_floor']:.3f}")
    print(f"   -> beyond 4.26 per 4-cycle-class instruction the step's {STEP_FULL} VOP2 instructions cost {s['vop2_cycles_implied']:.2f} cycles each on average (2.13 paired, ~4.2 alone)")
    k0 = np.unique(simd_key)[0]
    o = np.where(simd_key == k0)[0]
    o = o[np.argsort(t[o, 0])]
    base = t[o, 0].min()
    print("one SIMD, its waves in start order (M cycles from the SIMD's first entry) -- oldest first: a ladder completes every cadence, whatever is resident:")
    for w in o:
        print(f"   wave {w:5d} slot {hw['slot'][w]}  start {(t[w, 0] - base) / 1e6:7.3f}  ladder done {(t[w, 1] - base) / 1e6:7.3f}  exit {(t[w, 5] - base) / 1e6:7.3f}")
    if fused:
        is_inv = inv > 1000
        tot = (t[:, 5] - t[:, 0]).sum()
        print("share of all wave-cycles: ladder %.4f, first barrier wait %.4f, inversion (one wave per workgroup) %.4f, waiting for it %.4f, finish %.4f"
              % (ladder.sum() / tot, wait1.sum() / tot, inv[is_inv].sum() / tot, (wait2.sum() + inv[~is_inv].sum()) / tot, fin.sum() / tot))
        print(f"inversion phase of a workgroup: median {np.median(inv[is_inv]):.0f} cycles = {np.median(inv[is_inv]) / np.median(cad):.4f} of a ladder's SIMD time; "
              f"inverting waves by SIMD id: {dict(collections.Counter(hw['simd'][is_inv].tolist()))}")
    # per CU: cycles in which k waves are in their ladder phase (sweep over start/end events)
    hist = collections.Counter()
    for k in np.unique(cu_key):
        m = cu_key == k
        ev = sorted([(int(a), 1) for a in t[m, 0]] + [(int(b), -1) for b in t[m, 1]])
        cur, last = 0, ev[0][0]
        for when, d in ev:
            hist[cur] += when - last
            last, cur = when, cur + d
    tot_cu = sum(hist.values())
    print("per CU, share of its span with k waves in the ladder phase:  " + "  ".join(f"k={k}: {v / tot_cu:.4f}" for k, v in sorted(hist.items()) if v / tot_cu > 5e-4))
    return s


def section_models(asm_path):
    """instruction classes between consecutive s_memtime marks of the ladder loop of the level-2 build"""
    text = open(asm_path).read()
    _, lines = isa_mix.kernel_body(text, "k_x25519_fusedILb0ELi512")
    insts = [l.split(";")[0].strip() for l in lines if l.startswith("\t") and not l.strip().startswith(".")]
    insts = [i for i in insts if i]
    marks = [i for i, x in enumerate(insts) if x.startswith("s_memtime")]
    # marks in program order: the entry stamp, then the loop's eleven (step entry + ten sections), then the phase stamps
    assert len(marks) >= 12, marks
    s = 1
    out = []
    for q in range(10):
        seg = insts[marks[s + q] + 1: marks[s + q + 1]]
        c = collections.Counter(x.split()[0] for x in seg)
        valu = sum(v for k, v in c.items() if k.startswith("v_"))
        mad = c.get("v_mad_u64_u32", 0)
        full = sum(v for k, v in c.items() if k in isa_mix.FULL or k.replace("_e32", "") in isa_mix.FULL)
        half = valu - mad - full
        out.append((mad, half, full, len(seg) - valu))
    return out


NAMES = ["4 add/sub + 2 selects", "mul (x1-z1)(x2+z2)", "mul (x2-z2)(x1+z1)", "add + sub", "2 sqr", "mul by x1", "2 sqr", "mul x4", "sub + a24 step", "mul z4"]


def report_sections(ms, rec, models):
    sec = rec[:, 7:17].astype(np.float64)
    per = np.median(sec / sec.sum(axis=1, keepdims=True), axis=0)          # each section's share of a wave's step (wall)
    print(f"\n== sections of a ladder step (level-2 build of k_x25519_fused, {ms:.3f} ms: the marks cost time themselves): share of a step's wall time")
    print(f"{'section':28s} {'mad':>4s} {'half':>5s} {'full':>5s} {'other':>5s} {'model share':>12s} {'measured share':>15s} {'ratio':>6s}")
    cost = [4.26 * (m[0] + m[1]) + 2.7 * m[2] for m in models]
    for q in range(10):
        mad, half, full, other = models[q]
        print(f"{NAMES[q]:28s} {mad:4d} {half:5d} {full:5d} {other:5d} {cost[q] / sum(cost):12.4f} {per[q]:15.4f} {per[q] / (cost[q] / sum(cost)):6.3f}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("lib")
    ap.add_argument("--sections", nargs=2, metavar=("LIB2", "ASM2"))
    ap.add_argument("--n", type=int, default=1 << 20)
    ap.add_argument("--reps", type=int, default=6)
    ap.add_argument("--fused", action="store_true", help="probe k_x25519_fused (one launch) instead of the shipped two-launch shape")
    ap.add_argument("--dump", help="write the raw stamps here (.npz)")
    ap.add_argument("--json", help="write summary() here")
    args = ap.parse_args()
    print(f"# tools/cycle_probe.py {' '.join(sys.argv[1:])}")
    if args.lib.endswith(".npz"):                              # offline: analyse a dump of an earlier run
        d = np.load(args.lib)
        ms, rec = float(d["ms"]), d["rec"]
    else:
        ms, rec = measure(args.lib, args.n, args.fused, args.reps, verbose=True)
    if args.dump:
        np.savez_compressed(args.dump, ms=ms, rec=rec)
    s = report_phases(ms, rec, args.fused)
    if args.json:
        import json
        json.dump(s, open(args.json, "w"), indent=1)
    if args.sections:
        ms2, rec2 = measure(args.sections[0], args.n, True, args.reps, verbose=True)
        report_sections(ms2, rec2, section_models(args.sections[1]))


if __name__ == "__main__":
    main()
