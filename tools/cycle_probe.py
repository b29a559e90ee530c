#!/usr/bin/env python3
"""tools/cycle_probe.py -- in-kernel s_memtime measurement of k_x25519_fused on an UN-PROFILED run.

    python tools/cycle_probe.py build_ab/probe1.so [--sections build_ab/probe2.so build_ab/probe2.s] [--n 1048576]

probe1.so / probe2.so are the engine built with -DC25519_CYCLE_PROBE=1 / =2 (tools/build_variants.sh): every wave stamps
s_memtime (one tick = one shader cycle, MI355X_MICROARCH.md) at its phase boundaries and records the hardware slot it ran
on; level 2 also accumulates the ten sections of a ladder step.  Printed:
  * the sustained shader clock of the un-profiled kernel = (last exit - first entry, per CU) / HIP-event time;
  * SIMD cycles per ladder step per wave (the cadence at which a SIMD completes ladders / the steps of a ladder)
    against the issue model of the instruction stream (tools/isa_mix.py classes, nominal and measured cycle costs);
  * where the waves' time goes: ladder, barrier wait, the shared inversion, second barrier wait, finish;
  * with --sections: measured cycles of each section of the step against the model of the instructions between its marks.
"""
import argparse
import collections
import ctypes as C
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402


sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import isa_mix  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("lib")
ap.add_argument("--sections", nargs=2, metavar=("LIB2", "ASM2"))
ap.add_argument("--n", type=int, default=1 << 20)
ap.add_argument("--reps", type=int, default=6)
ap.add_argument("--split", action="store_true", help="probe k_x25519_ladder (the two-launch shape) instead of the fused kernel")
ap.add_argument("--dump", help="write the raw stamps here (.npz)")
args = ap.parse_args()
n = args.n
vp, sz = C.c_void_p, C.c_size_t
STEPS = 251 + 4 * 0.55            # 251 ladder steps + the opening doubling and the three closing ones (a doubling = 0.55 step)
# class costs in SIMD cycles per wave-instruction: nominal, and as measured in place by tools/ubench/mad_peak at four
# resident waves (profiles/r04_mad_peak.txt): MAD and the other 4-cycle-class instructions 4.26; a VOP2 instruction 2.13
# in a run of its own kind on all waves, but ~4 when it stands alone between MADs
MODELS = {"nominal": (4.0, 2.0), "measured, VOP2 always paired": (4.26, 2.13), "measured, VOP2 never paired": (4.26, 4.26)}


def load(path):
    L = C.CDLL(os.path.abspath(path))
    L.curve25519_dh_CreateSharedKey_dev.argtypes = [vp, vp, vp, sz, vp]
    L.c25519_amd_probe_set.argtypes = [vp]
    return L


def run(L, words):
    import torch
    from curve25519_amd import synth
    dev = torch.device("cuda", 0)
    sk_np, pk_np = synth.x25519_inputs(n)
    sk, pk = torch.from_numpy(sk_np).to(dev), torch.from_numpy(pk_np).to(dev)
    out = torch.empty((n, 32), dtype=torch.uint8, device=dev)
    waves = (n + 63) // 64
    buf = torch.zeros((waves, words), dtype=torch.int64, device=dev)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    os.environ["C25519_AMD_XF_SPLIT"] = "1" if args.split else "0"
    assert L.c25519_amd_probe_set(p(buf)) == 0
    best = None
    for r in range(args.reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        assert L.curve25519_dh_CreateSharedKey_dev(p(out), p(pk), p(sk), n, st) == 0     # keeps the clocks up
        a.record()
        assert L.curve25519_dh_CreateSharedKey_dev(p(out), p(pk), p(sk), n, st) == 0
        b.record()
        torch.cuda.synchronize()
        ms = a.elapsed_time(b)
        rec = buf.cpu().numpy().astype(np.uint64)
        if best is None or ms < best[0]:
            best = (ms, rec)
        print(f"  run {r}: {ms:.3f} ms")
    L.c25519_amd_probe_set(None)
    return best


def decode_hw(x):
    hw, xcc = x & np.uint64(0xffffffff), (x >> np.uint64(32)) & np.uint64(0xf)
    f = lambda lo, w: ((hw >> np.uint64(lo)) & np.uint64((1 << w) - 1)).astype(np.int64)  # noqa: E731
    return dict(slot=f(0, 4), simd=f(4, 2), cu=f(8, 4), sh=f(12, 1), se=f(13, 3), xcc=xcc.astype(np.int64))


def report_phases(ms, rec):
    t = rec[:, :6].astype(np.int64)
    hw = decode_hw(rec[:, 6])
    cu_key = ((hw["xcc"] * 8 + hw["se"]) * 2 + hw["sh"]) * 16 + hw["cu"]
    simd_key = cu_key * 4 + hw["simd"]
    print(f"\n== k_x25519_fused, n = {n}, un-profiled: {ms:.3f} ms (HIP events), {len(t)} waves on {len(set(cu_key))} CUs / {len(set(simd_key))} SIMDs")
    # s_memtime counters of different shader engines are not synchronised with each other: spans are taken per CU
    spans = np.array([int(t[cu_key == k, 5].max() - t[cu_key == k, 0].min()) for k in sorted(set(cu_key))])
    span = float(np.median(spans))
    clock = span / (ms * 1e-3) / 1e9
    print(f"kernel span per CU in shader cycles: min {spans.min() / 1e6:.3f} M, median {span / 1e6:.3f} M, max {spans.max() / 1e6:.3f} M")
    print(f"sustained shader clock of the un-profiled launch: {span / 1e6:.3f} M cycles / {ms:.3f} ms = {clock:.3f} GHz"
          + ("   (the HIP-event time includes k_batch_invert: a lower bound)" if args.split else ""))
    ladder, wait1, inv, wait2, fin = t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2], t[:, 4] - t[:, 3], t[:, 5] - t[:, 4]
    # A SIMD serves its resident waves oldest first: ladders complete one after the other at a fixed cadence, whatever
    # the number of resident waves.  The cadence (time between consecutive ladder completions on one SIMD, the first
    # four left out) is the SIMD time one ladder costs.
    cad = []
    for k in sorted(set(simd_key)):
        e = np.sort(t[simd_key == k, 1])
        cad.append(np.diff(e)[4:])
    cad = np.concatenate(cad)
    per_step = float(np.median(cad)) / STEPS
    print(f"ladder phase of a wave (wall): median {np.median(ladder) / 1e6:.3f} M cycles; ladder completions on one SIMD are {np.median(cad) / 1e6:.4f} M cycles apart"
          f" (p10 {np.percentile(cad, 10) / 1e6:.4f}, p90 {np.percentile(cad, 90) / 1e6:.4f})")
    print(f"SIMD cycles per ladder step per wave: {per_step:.0f}" + ("" if args.split else "   (fused: includes the inversion's share of the SIMD)"))
    for name, (cm, cf) in MODELS.items():
        model = (739 + 188) * cm + 319 * cf
        print(f"   issue model [{name}: MAD / half-rate class {cm:.2f}, VOP2 {cf:.2f} cycles]: {model:.0f} -> issue_model_frac {model / per_step:.3f}")
    k0 = sorted(set(simd_key))[0]
    o = np.where(simd_key == k0)[0]
    o = o[np.argsort(t[o, 0])]
    base = t[o, 0].min()
    print("one SIMD, its 16 waves in start order (M cycles from the SIMD's first entry):")
    for w in o:
        print(f"   wave {w:5d} slot {hw['slot'][w]}  start {(t[w, 0] - base) / 1e6:7.3f}  ladder done {(t[w, 1] - base) / 1e6:7.3f}  exit {(t[w, 5] - base) / 1e6:7.3f}")
    is_inv = inv > 1000
    tot = (t[:, 5] - t[:, 0]).sum()
    print("share of all wave-cycles: ladder %.4f, first barrier wait %.4f, inversion (one wave per workgroup) %.4f, waiting for it %.4f, finish %.4f"
          % (ladder.sum() / tot, wait1.sum() / tot, inv[is_inv].sum() / tot, (wait2.sum() + inv[~is_inv].sum()) / tot, fin.sum() / tot))
    print(f"inversion phase of a workgroup: median {np.median(inv[is_inv]):.0f} cycles = {np.median(inv[is_inv]) / np.median(ladder):.4f} of a ladder phase")
    # per CU: cycles in which k waves are in their ladder phase (sweep over start/end events)
    hist = collections.Counter()
    busy_any = 0
    for k in sorted(set(cu_key)):
        m = cu_key == k
        ev = sorted([(int(a), 1) for a in t[m, 0]] + [(int(b), -1) for b in t[m, 1]])
        cur, last = 0, ev[0][0]
        for when, d in ev:
            hist[cur] += when - last
            last, cur = when, cur + d
        busy_any += int(t[m, 5].max() - t[m, 0].min())
    tot_cu = sum(hist.values())
    print("per CU, share of its span with k waves in the ladder phase:  " + "  ".join(f"k={k}: {v / tot_cu:.4f}" for k, v in sorted(hist.items()) if v / tot_cu > 5e-4))
    full = max(hist)
    print(f"  -> a CU holds its full {full} ladder waves {hist[full] / tot_cu:.4f} of the time; weighted ladder occupancy {sum(k * v for k, v in hist.items()) / tot_cu / full:.4f}")
    return clock, per_step


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
    print(f"\n== sections of a ladder step (level-2 build, {ms:.3f} ms: the marks cost time themselves): share of a step's wall time")
    print(f"{'section':28s} {'mad':>4s} {'half':>5s} {'full':>5s} {'other':>5s} {'model share':>12s} {'measured share':>15s} {'ratio':>6s}")
    cost = [4.26 * (m[0] + m[1]) + 2.7 * m[2] for m in models]
    for q in range(10):
        mad, half, full, other = models[q]
        print(f"{NAMES[q]:28s} {mad:4d} {half:5d} {full:5d} {other:5d} {cost[q] / sum(cost):12.4f} {per[q]:15.4f} {per[q] / (cost[q] / sum(cost)):6.3f}")


print(f"# tools/cycle_probe.py {' '.join(sys.argv[1:])}")
if args.lib.endswith(".npz"):                              # offline: analyse a dump of an earlier run
    d = np.load(args.lib)
    ms, rec = float(d["ms"]), d["rec"]
else:
    L1 = load(args.lib)
    words = L1.c25519_amd_probe_words()
    ms, rec = run(L1, words)
if args.dump:
    np.savez_compressed(args.dump, ms=ms, rec=rec)
report_phases(ms, rec)
if args.sections:
    L2 = load(args.sections[0])
    ms2, rec2 = run(L2, words)
    report_sections(ms2, rec2, section_models(args.sections[1]))
