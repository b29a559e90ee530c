#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload x25519|mixed|sign|verify]

With --gpus N > 1 and no WORLD_SIZE in the environment the script re-launches itself under
`python -m torch.distributed.run --nproc-per-node N` (one rank per GPU); launched by torchrun it reads
RANK / LOCAL_RANK / WORLD_SIZE as usual.  Rank 0 prints ONE JSON line.

A "step" is one pass of the hot path over one batch of 2^20 operations per GPU, inputs already resident in HBM,
followed for N > 1 by the single RCCL gather of the result rows to rank 0 that north_star names (asynchronous,
overlapped with the next batch; every gather of the timed steps completes inside the timed region).  Weak scaling:
every rank owns its own 2^20 operations.

Timing protocol of every block (run_timed): W warm-up steps; barrier; clock-ramp launches of the same pass (the ranks
left the barrier together, whatever each did before it); synchronize; t0; EXACTLY K steps; every gather of those steps
complete; synchronize; t1; barrier.  A rank's time is its own t1 - t0 (device events between the same two points ride
along as a cross-check), the block's time the MAX over ranks.  After the timed steps every rank hashes its last output
buffer (SHA-256) against the committed digest of the reference's portable-C outputs for its rank's inputs
(tests/golden/digests.json, written by tests/golden/gen_golden.py from oracle/_ref): `bit_exact`; a mismatch makes the
run exit non-zero behind the JSON line.

BASELINE.json's metric is a pair -- "X25519 shared-key ops/sec + Ed25519 verifies/sec" -- so the default run times
BOTH, each with its own W warm-up + K timed steps (mean of steps, max over ranks):
  value / ms_per_step / roofline  -- X25519 curve25519_dh_CreateSharedKey (configs[1]); `value` is this number;
  verify                          -- ed25519_VerifySignature on configs[3]'s set (valid + 1/64 corrupted);
  sign                            -- ed25519_SignMessage (configs[2]);
roofline.verify / roofline.sign repeat the two side results inside the contract's roofline object.
--workload mixed times BASELINE.json configs[4]: X25519 + sign + verify by contiguous thirds of each GPU's 2^20,
one gather per output type.  Each roofline carries the HBM fraction the contract asks for (tiny by construction)
and `valu`: algorithmic 32x32 MACs against the measured v_mad_u64_u32 issue peak, the roof that binds, plus -- measured
live on an un-profiled launch of the probe build (s_memtime inside the kernel) -- the SIMD cycles one ladder step costs
against the issue model of its instruction stream (`issue_model_frac`) and the shader clock it ran at.
cpu_baseline: the reference's portable-C path (oracle/_ref) or the oracle port on this host's cores.
"""
import argparse
import contextlib
import glob
import json
import os
import socket
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BATCH = 1 << 20
BYTES_PER_OP = {"x25519": 96, "sign": 160, "verify": 132}          # SURVEY.md 8(d), compulsory HBM bytes
MACS_PER_OP = {"x25519": 184104, "sign": 52992, "verify": 245664}  # SURVEY.md 8(a), 32x32 MACs at 72/mul
# what the device actually issues per operation: profiles/rNN_executed_macs.json -- v_mad_u64_u32 / v_mad_i64_i32 instructions
# counted by running the DEVICE SOURCE one lane at a time against the C model of the gfx950 primitives (tools/executed_macs.py;
# tests/test_bench_contract.py pins the file against a fresh count).  No committed count: the line says so (null), it does not
# fall back on constants.
EXECUTED_OPS = ("x25519", "sign", "verify")


def executed_macs():
    path = latest_profile("r[0-9][0-9]_executed_macs.json")
    if path:
        try:
            with open(path) as f:
                d = json.load(f)["per_op"]
            return {k: int(d[k]) for k in EXECUTED_OPS}, os.path.basename(path)
        except Exception:
            pass
    return {k: None for k in EXECUTED_OPS}, None

HBM_PEAK_GBS = 8000.0                                               # MI355X_MICROARCH.md

PASS_KERNELS = {
    "x25519": ("k_x25519_ladder", "k_batch_invert<FinishX25519>"),     # batches above 2^16 (k_x25519_fused below)
    "sign": ("k_ed25519_sign_mult<false, true>", "k_batch_invert<FinishPack>", "k_ed25519_sign_finish"),
    "verify": ("k_ed25519_verify_fast_scalars", "k_ed25519_verify_fast_points", "k_ed25519_verify_fast_walk",
               "k_ed25519_verify_slow"),                # (the slow list is empty for on-curve keys: a ~10 us launch)
}
METRIC_NAME = {
    "x25519": "X25519 shared-key ops/sec (batch=2^20 per GPU, variable-base Montgomery ladder) [+ Ed25519 verifies/sec "
              "in `verify`]",
    "sign": "Ed25519 signs/sec (batch=2^20 per GPU, fixed-base signed-comb walk, 32-byte messages)",
    "verify": "Ed25519 verifies/sec (batch=2^20 per GPU, distinct keys; exact lattice-shortened double-scalar walk for "
              "on-curve keys, the reference's 4-fold + 8-fold order otherwise)",
    "mixed": "mixed X25519 + Ed25519 sign + verify ops/sec (contiguous thirds of 2^20 per GPU)",
}
WORKLOAD_NAME = {
    "x25519": "BASELINE.json configs[1]: batch 2^20 X25519 curve25519_dh_CreateSharedKey per GPU, one keypair per "
              "lane, inputs resident in HBM",
    "sign": "BASELINE.json configs[2]: batch 2^20 ed25519_SignMessage per GPU (shipped: the 13 x 20 signed comb read through L2; "
            "the 8 x 32 comb staged in LDS the config names is timed in extra.sign_lds_comb_per_s)",
    "verify": "BASELINE.json configs[3]: batch 2^20 ed25519_VerifySignature per GPU (Verify_Init + Verify_Check), "
              "config-3 signatures with the seeded 1/64 corrupted entries",
    "mixed": "BASELINE.json configs[4]: 2^20 per GPU (2^23 over 8) as contiguous thirds X25519 / sign / verify, "
             "one RCCL gather per output type",
}


CLOCK_RAMP_S = 0.06      # untimed launches behind the opening barrier, directly in front of every block's timed steps (run_timed)
DIGESTS = os.environ.get("C25519_BENCH_DIGESTS") or os.path.join(ROOT, "tests", "golden", "digests.json")   # (the env: tests only)
DIGEST_KEY = {"x25519": "x25519_shared", "sign": "ed25519_sig", "verify": "ed25519_verdicts"}


def expected_digests(rank, world, n):
    """The committed SHA-256 digests of the REFERENCE's outputs (portable-C build, oracle/_ref; tests/golden/gen_golden.py)
    for rank `rank`'s n elements -- rank r of a world > 1 draws its inputs from seed + 0x100 * r (synth.rank_seed_shift); the
    seeded streams are positional, so a power-of-two batch below 2^20 is a prefix of the 2^20 one.  None: no digest
    committed for this rank / size (the line then says bit_exact: null, it does not guess)."""
    try:
        with open(DIGESTS) as f:
            d = json.load(f)
        rec = d["ranks"]["by_rank"][rank if world > 1 else 0]
        return rec["prefix"].get(str(n))
    except Exception:
        return None



def latest_profile(pattern):
    hits = sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)))
    return hits[-1] if hits else None


def measured_mad_peak():
    """lane-MAC/s of v_mad_u64_u32 in its ACCUMULATING form (d = a*b + d, what a multi-precision column is made of) on
    MI355X at 8 waves per SIMD: tools/ubench/mad_peak -- a whole-asm loop of 128 MADs per trip over eight independent
    chains (committed summary).  It is the half-rate class' ~37.5 T.  The compiler-generated 16-per-trip loops of
    tools/ubench/valu_rates, which rounds 1 and early 2 divided by, read 28-32 T: loop overhead, fetch-line placement
    of the loop top and s_nop padding between asm statements, not the instruction (profiles/README.md)."""
    best, src = None, None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_mad_peak.json"))):
        try:
            with open(path) as f:
                r = json.load(f)["rates"]["v_mad_u64_u32"]
        except Exception:
            continue
        if best is None or r > best:                 # boxes differ in sustained clock: the fastest one seen is the roof
            best, src = r, os.path.basename(path)
    return best, src


def live_mad_peak():
    """The same accumulating v_mad_u64_u32 stream as measured_mad_peak(), measured NOW on this device: tools/ubench/mad_peak --quick
    (one configuration, ~60 ms of clock ramp + 10 launches of ~1 ms), run by rank 0 behind every timed block.  The denominator
    of roofline.*.valu when it is there: the numerator's box, clock and run."""
    import subprocess
    exe = os.path.join(ROOT, "tools", "ubench", "mad_peak")
    if not os.path.exists(exe):
        return None
    try:
        out = subprocess.run([exe, "--quick"], capture_output=True, text=True, timeout=120)
        rec = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
        return rec if rec.get("v_mad_u64_u32") else None
    except Exception as e:                                   # a measurement leg must not take the bench line down
        print(f"bench.py: tools/ubench/mad_peak --quick failed ({e!r}); the committed peak stays the denominator", file=sys.stderr)
        return None


def apply_live_peak(valu, live):
    """valu.peak / frac / frac_executed against the live peak (peak_live); the committed figure stays beside it (peak_committed)."""
    valu["peak_committed"] = valu.get("peak")
    valu["peak_committed_source"] = valu.pop("peak_source", None)
    valu["peak_committed_policy"] = valu.pop("peak_policy", None)
    valu["frac_vs_committed_peak"] = valu.get("frac")
    if not live:
        valu["peak_live"] = None
        valu["peak_policy"] = "committed (no live measurement in this run): " + str(valu["peak_committed_policy"])
        return
    peak = live["v_mad_u64_u32"]
    valu["peak_live"] = round(peak / 1e12, 4)
    valu["peak"] = valu["peak_live"]
    valu["frac"] = round(valu["achieved"] * 1e12 / peak, 4)
    if valu.get("executed_macs_per_op") and valu.get("algorithmic_macs_per_op"):
        valu["frac_executed"] = round(valu["achieved"] * 1e12 * valu["executed_macs_per_op"] / valu["algorithmic_macs_per_op"] / peak, 4)
    valu["peak_policy"] = ("live: tools/ubench/mad_peak --quick on this device behind this run's timed blocks (the 8-waves-per-SIMD "
                           "accumulating v_mad_u64_u32 stream, best of 10 launches after a 60 ms ramp)")


def valu_issue(kernels):
    """How busy the vector ALUs are in every kernel of a pass, from the committed counter passes of this bench: VALU-busy against
    chip peak (north_star) and the issue utilisation by instruction class (tools/valu_issue.py)."""
    try:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import valu_issue as vi
        return vi.for_pass(kernels)
    except Exception as e:
        print(f"bench.py: tools/valu_issue.py failed ({e!r})", file=sys.stderr)
        return None


def single_call_latencies(L, e, reps=300):
    """ONE operation through the reference's own prototypes (include/curve25519_dh.h, include/ed25519_signature.h: host pointers, a
    device batch of one), wall clock per call in microseconds, outputs checked against the device-resident batch's"""
    import ctypes as C
    buf = lambda b: (C.c_ubyte * len(b)).from_buffer_copy(bytes(b))  # noqa: E731
    row = lambda t: t[0].cpu().numpy().tobytes()  # noqa: E731
    esk, msg, pub_w, priv_w, sig_w = row(e["esk"]), row(e["msg"]), row(e["pub"]), row(e["priv"]), row(e["sig"])
    pub, priv, sig, shared = (C.c_ubyte * 32)(), (C.c_ubyte * 64)(), (C.c_ubyte * 64)(), (C.c_ubyte * 32)()
    sk, pk = buf(esk), buf(pub_w)

    def wall(fn):
        for _ in range(30):
            fn()
        t = time.perf_counter()
        for _ in range(reps):
            fn()
        return round((time.perf_counter() - t) / reps * 1e6, 1)

    out = {"curve25519_dh_CreateSharedKey": wall(lambda: L.curve25519_dh_CreateSharedKey(shared, pk, sk)),
           "ed25519_CreateKeyPair": wall(lambda: L.ed25519_CreateKeyPair(pub, priv, None, buf(esk))),
           "ed25519_SignMessage": wall(lambda: L.ed25519_SignMessage(sig, priv, None, buf(msg), len(msg))),
           "ed25519_VerifySignature": wall(lambda: L.ed25519_VerifySignature(sig, pub, buf(msg), len(msg)))}
    out["bytes_equal_the_batch"] = bool(bytes(pub) == pub_w and bytes(priv) == priv_w and bytes(sig) == sig_w
                                        and L.ed25519_VerifySignature(sig, pub, buf(msg), len(msg)) == 1)
    return out


def measured_traffic(kernels):
    """HBM bytes per pass from the committed rocprofv3 PMC passes (separate --pmc runs of this same bench,
    summarised by tools/rocpd_summary.py): WRITE_SIZE + 2 x FETCH_SIZE, both in KiB -- the x2 is the gfx950
    FETCH_SIZE correction of MI355X_MICROARCH.md (HBM section)."""
    try:
        path = latest_profile("r[0-9][0-9]_pmc.json")
        with open(path) as f:
            d = json.load(f)
        def rec(name):                      # rocprof prints template kernels as "void name<...>"
            hits = [v for k, v in d.items() if name.rstrip(">") in k and "<true>" not in k]   # "<FinishPack>" matches "<FinishPack, 16>"
            return hits[0]
        total = sum((2.0 * rec(k)["FETCH_SIZE"] + rec(k)["WRITE_SIZE"]) * 1024 for k in kernels)
        return int(total), os.path.basename(path)
    except Exception:
        return None, None


def issue_model(n, live):
    """SIMD cycles per ladder step of the X25519 ladder kernel, measured by the s_memtime stamps of the probe build
    (curve25519_amd/libcurve25519_amd_probe.so, `python -m curve25519_amd.build --probe`) on an un-profiled launch of
    this very batch, against the issue model of the step's instruction stream: sum over instruction classes of count x
    cycle cost (tools/cycle_probe.py; class costs measured by tools/ubench/mad_peak).  live=False, or no probe build:
    the committed measurement of the round."""
    from curve25519_amd import build as _b
    if live and os.path.exists(_b.PROBE_LIB):
        try:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import cycle_probe
            ms, rec = cycle_probe.measure(_b.PROBE_LIB, n, fused=n <= (1 << 16), reps=3)
            out = cycle_probe.summary(ms, rec)[0]
            out["source"] = "live: s_memtime stamps of libcurve25519_amd_probe.so, this run"
            return out
        except Exception as e:                                   # a measurement leg must not take the bench line down
            print(f"bench.py: cycle probe failed ({e!r}); falling back to the committed measurement", file=sys.stderr)
    path = latest_profile("r[0-9][0-9]_cycle_probe.json")
    if not path:
        return None
    with open(path) as f:
        out = json.load(f)
    out["source"] = f"profiles/{os.path.basename(path)} (committed measurement, not this run)"
    return out


def roofline_for(wl, n, kernel_ms, probe=None):
    """The contract's roofline object for one pass of workload `wl` (n operations, mean kernel time kernel_ms).
    probe: issue_model()'s result for the X25519 pass."""
    kernel_s = kernel_ms * 1e-3
    achieved_gbs = BYTES_PER_OP[wl] * n / kernel_s / 1e9
    peak_mac, peak_src = measured_mad_peak()
    kernels = PASS_KERNELS[wl] if wl != "x25519" or n > (1 << 16) else ("k_x25519_fused",)
    traffic, traffic_src = measured_traffic(kernels)
    achieved_mac = MACS_PER_OP[wl] * n / kernel_s
    executed, executed_src = executed_macs()
    valu = {"bound": "valu v_mad_u64_u32 issue", "achieved": round(achieved_mac / 1e12, 4),
            "peak": round(peak_mac / 1e12, 4) if peak_mac else None, "unit": "T 32x32 MAC/s",
            "frac": round(achieved_mac / peak_mac, 4) if peak_mac else None,
            "algorithmic_macs_per_op": MACS_PER_OP[wl],
            "executed_macs_per_op": executed[wl],
            "executed_macs_source": f"profiles/{executed_src} (device source run against the C model of the primitives, "
                                    "tools/executed_macs.py)" if executed_src else None,
            "frac_executed": round(executed[wl] * n / kernel_s / peak_mac, 4) if peak_mac and executed[wl] else None,
            "issue": valu_issue(kernels),
            "peak_source": f"profiles/{peak_src}" if peak_src else None,
            "peak_policy": "the fastest box's 8-waves-per-SIMD v_mad_u64_u32 stream of all committed rounds (wall-clock rate)"}
    if probe:
        attach_probe(valu, probe)
    return finish_roofline(wl, n, kernel_ms, achieved_gbs, traffic, traffic_src, kernels, valu)


def attach_probe(valu, probe):
    """issue_model()'s result into a roofline's `valu` object: the fraction that says how close the kernel is to what its
    instruction stream allows -- class-cost cycles of one ladder step / SIMD cycles measured per step (un-profiled,
    in-kernel).  Headline = the floor of this stream at four waves: 4.26 cycles per 4-cycle-class instruction, every VOP2
    instruction paired with another wave's and executed inside the former's issue bubbles as far as those reach
    (tools/cycle_probe.py: model_cycles("floor"))."""
    valu.update({"issue_model_frac": probe.get("issue_model_frac_floor", probe.get("issue_model_frac_measured_vop2_paired")),
                 "issue_model_cycles_per_step": probe.get("issue_model_cycles_floor"),
                 "issue_model_frac_nominal_4_4_2": probe.get("issue_model_frac_nominal"),
                 "simd_cycles_per_ladder_step": probe.get("simd_cycles_per_ladder_step"),
                 "ladder_step_instructions": probe.get("ladder_step_instructions"),
                 "vop2_cycles_implied": probe.get("vop2_cycles_implied"),
                 "shader_clock_GHz": probe.get("shader_clock_GHz"),
                 "issue_model_source": probe.get("source")})


def binding_frac(valu):
    """The fraction of the binding (v_mad_u64_u32 issue) roof: the reference's operation count over time -- unless the
    device does LESS work than the reference (sign's and verify's re-designed walks), where that quotient is a speed-up,
    not a utilisation, and the executed count is the honest one."""
    fr = [f for f in (valu.get("frac"), valu.get("frac_executed")) if f is not None]
    return min(fr) if fr else None


def finish_roofline(wl, n, kernel_ms, achieved_gbs, traffic, traffic_src, kernels, valu):
    return {
        "bound": "hbm", "achieved": round(achieved_gbs, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": round(achieved_gbs / HBM_PEAK_GBS, 6), "traffic": traffic,
        "traffic_source": f"profiles/{traffic_src}: (2*FETCH_SIZE + WRITE_SIZE) KiB per pass" if traffic_src else None,
        "kernel": " + ".join(kernels), "kernel_ms": round(kernel_ms, 4),
        "algorithmic_bytes_per_launch": BYTES_PER_OP[wl] * n,
        "traffic_over_algorithmic": round(traffic / (BYTES_PER_OP[wl] * n), 2) if traffic else None,
        # the contract's top-level fields are the HBM roof (bound = "hbm"); the roof that BINDS this path is the integer
        # multiplier: `binding` names it and `binding_frac` is that roof's fraction (= valu.frac)
        "binding": "valu", "binding_frac": binding_frac(valu),
        "note": "VALU-integer bound path: the HBM fraction is tiny by construction, `valu` (v_mad_u64_u32 issue) is the roof "
                "that binds: read binding_frac / valu.frac, not frac." + (
                    "  X25519 above 2^16 runs as two launches (ladder, shared inversion): the projective results cross HBM "
                    "once each way, which is the measured traffic over the algorithmic 96 B/op -- 0.04 ms of a 7.9 ms pass."
                    if wl == "x25519" and n > (1 << 16) else ""),
        "valu": valu,
    }


def cpu_baseline(quick=False):
    """Time curve25519_dh_CreateSharedKey (plus Ed25519 sign / verify) on the host cores: the real reference
    (portable-C build, oracle/_ref) when its prebuilt library travelled with the repo, else the oracle port.
    A C thread pool fans one contiguous slice per core; bounded sample, same input distribution as the GPU
    workload.  quick: the shorter sample used beside multi-GPU runs."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_lib import Oracle, Reference
    from curve25519_amd import synth
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:                                   # cgroup v2 CPU quota of the container, if any
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = float(q) / float(per)
            cores = max(1, min(cores, int(round(quota))))
    except Exception:
        pass
    budget_s = 3.0 if quick else 10.0
    n = (2048 if quick else 8192) * cores
    n_ed = cores * (1024 if quick else 4096)               # sign / verify samples of a few hundred ms, not 30 ms
    sk, pk = synth.x25519_inputs(n)
    esk, msg = synth.ed25519_inputs(n_ed)
    orc = Oracle()
    pub, priv = orc.ed25519_keypair(esk, threads=cores)
    if Reference.available():
        ref, kind = Reference(), "reference"
        shared = lambda p, s, t: ref.x25519_shared_threaded(p, s, t)          # noqa: E731
        sign = lambda t: ref.ed25519_sign_threaded(priv, msg, t)              # noqa: E731
        verify = lambda sg, t: ref.ed25519_verify_threaded(sg, pub, msg, t)   # noqa: E731
    else:
        kind = "port"
        shared = lambda p, s, t: orc.x25519_shared(p, s, threads=t)           # noqa: E731
        sign = lambda t: orc.ed25519_sign(priv, msg, threads=t)               # noqa: E731
        verify = lambda sg, t: orc.ed25519_verify(sg, pub, msg, threads=t)    # noqa: E731

    def np_equal(a, b):
        import numpy as np
        return np.array_equal(a, b)

    def rate(fn, count):
        t0 = time.perf_counter()
        out = fn()
        return count / (time.perf_counter() - t0), out

    n1 = 1024 if quick else 4096
    single, _ = rate(lambda: shared(pk[:n1], sk[:n1], 1), n1)              # config 1: sequential calls, one core
    # all cores, in rounds of 1024 per thread, until the sample is used up or the time budget has gone by
    done, t0, chunk = 0, time.perf_counter(), 1024 * cores
    while done < n and time.perf_counter() - t0 < budget_s:
        shared(pk[done:done + chunk], sk[done:done + chunk], cores)
        done += chunk
    multi, n = done / (time.perf_counter() - t0), done
    sign_rate, sig = rate(lambda: sign(cores), n_ed)
    verify_rate, ok = rate(lambda: verify(sig, cores), n_ed)
    asm = None
    if Reference.available(asm=True):                  # the reference's x86-64 assembly back-end (its README's headline path)
        ra = Reference(asm=True)
        a1, _ = rate(lambda: ra.x25519_shared_threaded(pk[:n1], sk[:n1], 1), n1)
        na = min(n, 4096 * cores)
        am, _ = rate(lambda: ra.x25519_shared_threaded(pk[:na], sk[:na], cores), na)
        asig_rate, asig = rate(lambda: ra.ed25519_sign_threaded(priv, msg, cores), n_ed)
        aver_rate, aok = rate(lambda: ra.ed25519_verify_threaded(asig, pub, msg, cores), n_ed)
        asm = {"kind": "reference, x86-64 assembly back-end (source/asm64; oracle/_ref/libcurve25519_ref_asm.so)",
               "value": round(am, 1), "unit": "X25519 shared-key ops/s", "cores": cores,
               "single_core_ops_per_s": round(a1, 1), "ed25519_sign_per_s": round(asig_rate, 1),
               "ed25519_verify_per_s": round(aver_rate, 1), "outputs_equal_portable_c": bool(np_equal(asig, sig) and aok.all())}
    model = ""
    try:
        with open("/proc/cpuinfo") as f:
            model = next(l.split(":", 1)[1].strip() for l in f if l.startswith("model name"))
    except Exception:
        pass
    return {"value": round(multi, 1), "unit": "X25519 shared-key ops/s", "cores": cores, "kind": kind,
            "sample": f"{n} curve25519_dh_CreateSharedKey calls over {cores} threads (C thread pool, one contiguous "
                      f"slice each), seeded uniform sk/pk; {n_ed} signs and {n_ed} verifies the same way",
            "single_core_ops_per_s": round(single, 1), "ed25519_sign_per_s": round(sign_rate, 1),
            "ed25519_verify_per_s": round(verify_rate, 1), "ed25519_verify_all_valid": bool(ok.all()),
            "host_logical_cpus": os.cpu_count(), "cgroup_cpu_quota": quota,
            "cpu_model": model, "asm_backend": asm}


def device_identity(torch, dev_index):
    """What this rank's device IS, for the line's `ranks` attestation: UUID, PCI address, name, compute units -- so that "did
    RCCL see N ranks on N distinct devices" is answerable from the record alone (rccl.h:745: one ncclGather, every peer on a
    device of its own)."""
    pr = torch.cuda.get_device_properties(dev_index)
    uuid = getattr(pr, "uuid", None)
    pci = None
    if hasattr(pr, "pci_bus_id"):
        pci = "%04x:%02x:%02x" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, getattr(pr, "pci_device_id", 0))
    return {"device_index": dev_index, "device_name": pr.name, "uuid": str(uuid) if uuid is not None else None,
            "pci_bus_id": pci, "compute_units": pr.multi_processor_count, "gcn_arch": getattr(pr, "gcnArchName", None),
            "hbm_bytes": pr.total_memory}


def attest_ranks(identities, world, backend, share_gpu):
    """The `rccl` object of the line from every rank's device_identity(): the ranks' devices must be pairwise distinct
    (UUID and PCI address) unless the shared-GPU launcher self-test knob is set -- a run whose ranks sat on one device is
    not a multi-GPU run whatever the launcher said."""
    keys = [(a.get("uuid"), a.get("pci_bus_id"), a.get("device_index")) for a in identities]
    hw = [(a.get("uuid"), a.get("pci_bus_id")) for a in identities]
    distinct = len(set(keys)) == len(keys)
    distinct_hw = len(set(hw)) == len(hw) and all(u is not None or p is not None for u, p in hw)
    pcis = [a.get("pci_bus_id") for a in identities]
    # CPX: the logical devices of ONE MI355X are distinct devices that share its PCI address
    one_package = world > 1 and distinct and pcis[0] is not None and len(set(pcis)) == 1
    return {"backend": backend, "world": world, "ranks_seen": len(identities),
            "compute_partitions_of_one_device": bool(one_package or any(a.get("compute_partition") for a in identities)),
            "devices_distinct": bool(distinct), "devices_distinct_by_uuid_or_pci": bool(distinct_hw),
            "shared_gpu_selftest": bool(share_gpu),
            "is_multi_gpu_measurement": bool(world > 1 and distinct and not share_gpu and backend == "nccl" and not one_package
                                             and not any(a.get("compute_partition") for a in identities))}


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: become `python -m torch.distributed.run ... bench.py ...`."""
    import torch
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if have < args.gpus and not (have and os.environ.get("C25519_BENCH_SHARE_GPU")):
        raise SystemExit(f"bench.py --gpus {args.gpus}: this machine exposes {have} GPU(s) to this process "
                         "(the launcher is fine; the devices are missing)")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    sys.stdout.flush()
    os.execv(sys.executable, cmd)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=BATCH, help="operations per GPU per step (default 2^20)")
    ap.add_argument("--no-side", "--no-extra", action="store_true", dest="no_side",
                    help="time the primary workload only (skip the verify / sign / keypair / one-key measurements)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline")
    ap.add_argument("--one-stream", action="store_true",
                    help="mixed workload: issue the three sub-batches on ONE stream (default: one stream each)")
    ap.add_argument("--workload", choices=("x25519", "sign", "verify", "mixed"), default="x25519",
                    help="x25519 = BASELINE.json configs[1] (the default and the driver's contract; also times verify and "
                         "sign); sign / verify = configs[2] / configs[3] alone; mixed = configs[4]")
    ap.add_argument("--dist-selftest", action="store_true",
                    help="run the N>1 code path (process group + RCCL gather) with a world of one rank")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)                              # does not return

    # stdout carries exactly ONE line, the result: whatever libraries print at C level (RCCL's version banner, ...)
    # is routed to stderr for the rest of the run
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    from curve25519_amd import synth
    from curve25519_amd.sharded import HipEngine, OverlappedGather, mixed_thirds

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher set WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: torch.cuda.is_available() is False (no CPU fallback)")
    # C25519_BENCH_SHARE_GPU=1: launcher self-test on a box with fewer GPUs than ranks -- the ranks share the devices
    # that exist and gather over gloo (RCCL refuses a duplicate device).  Exercises the self-launch, rank and gather
    # bookkeeping of the N > 1 path; the number it prints is NOT a multi-GPU measurement and says so.
    share_gpu = bool(os.environ.get("C25519_BENCH_SHARE_GPU")) and torch.cuda.device_count() < world
    dev_index = local_rank % torch.cuda.device_count() if share_gpu else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    use_dist = world > 1 or args.dist_selftest
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        if share_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    # ---- who is here: every rank's device, the collective backend and its version, gathered to rank 0 for the line ----
    ident = device_identity(torch, dev_index)
    ident.update({"rank": rank, "local_rank": local_rank, "hostname": socket.gethostname(), "pid": os.getpid(),
                  "backend": dist.get_backend() if use_dist else None,
                  "rccl_version": ".".join(str(v) for v in torch.cuda.nccl.version()) if hasattr(torch.cuda, "nccl") else None,
                  "compute_partition": os.environ.get("C25519_BENCH_PARTITION_NOTE")})
    identities = [ident]
    if use_dist:
        identities = [None] * world
        dist.all_gather_object(identities, ident)
        identities.sort(key=lambda a: a["rank"])
    rccl = attest_ranks(identities, world, dist.get_backend() if use_dist else None, share_gpu)
    if world > 1 and not share_gpu and not rccl["devices_distinct"]:
        raise SystemExit("bench.py: two ranks of this run sit on the same device (" + json.dumps(identities) + "): not a multi-GPU "
                         "run.  (C25519_BENCH_SHARE_GPU=1 is the launcher self-test that allows it, over gloo.)")

    n = args.batch
    eng = HipEngine(dev)
    seed_shift = synth.rank_seed_shift(rank, world)        # every rank owns its own 2^20 elements (weak scaling)
    delay_rank0_s = float(os.environ.get("C25519_BENCH_DELAY_RANK0_S", "0") or 0)   # test knob (run_timed)
    up = lambda a: torch.from_numpy(a).to(dev)             # noqa: E731

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- resident inputs and one launch closure per pass --------------------------------------------------
    def make_x25519(m, lo=0):
        sk = up(synth.random_bytes((n, 32), synth.SEED_X25519_SK + seed_shift)[lo:lo + m])
        pk = up(synth.random_bytes((n, 32), synth.SEED_X25519_PK + seed_shift)[lo:lo + m])
        return {"wl": "x25519", "n": m, "width": 32, "dtype": torch.uint8, "sk": sk,
                "launch": lambda dst: eng.api.curve25519_dh_CreateSharedKey_dev(dst, pk, sk)}

    ed_cache = {}

    def ed_material():
        if not ed_cache:
            esk = up(synth.random_bytes((n, 32), synth.SEED_ED_SK + seed_shift))
            msg_np = synth.random_bytes((n, 32), synth.SEED_ED_MSG + seed_shift)
            msg = up(msg_np)
            pub, priv = eng.ed25519_keypair(esk)
            sig = eng.ed25519_sign(priv, msg)
            vsig_np, vmsg_np, bad = synth.corrupt_for_verify(sig.cpu().numpy(), msg_np)
            ed_cache.update(esk=esk, msg=msg, pub=pub, priv=priv, sig=sig, vsig=up(vsig_np), vmsg=up(vmsg_np), bad=bad)
        return ed_cache

    def make_sign(m, lo=0):
        e = ed_material()
        priv, msg = e["priv"][lo:lo + m], e["msg"][lo:lo + m]
        return {"wl": "sign", "n": m, "width": 64, "dtype": torch.uint8,
                "launch": lambda dst: eng.api.ed25519_SignMessage_dev(dst, priv, msg)}

    def make_verify(m, lo=0):
        e = ed_material()
        vsig, pub, vmsg = e["vsig"][lo:lo + m], e["pub"][lo:lo + m], e["vmsg"][lo:lo + m]
        return {"wl": "verify", "n": m, "width": 1, "dtype": torch.int32, "bad": e["bad"][lo:lo + m],
                "launch": lambda dst: eng.api.ed25519_VerifySignature_dev(dst, vsig, pub, vmsg)}

    def digest_of(t):
        import hashlib
        a = t.detach().cpu().numpy()
        return hashlib.sha256(a.astype("<i4").tobytes() if a.dtype.itemsize == 4 else a.tobytes()).hexdigest()

    def run_timed(passes, ramp=True, mixed=False):
        """W warm-up + K timed steps of the given passes (each step launches every pass once, then starts its gathers).
        Returns a dict: elapsed (s, max over ranks), kernel_ms per pass, outs, attribution (per rank), bit_exact per pass."""
        outs = [torch.empty((p["n"], p["width"]), dtype=p["dtype"], device=dev) for p in passes]
        ogs = [OverlappedGather(p["n"], p["width"], dev, root=0, dtype=p["dtype"]) if use_dist else None for p in passes]

        # several passes per step (the mixed workload): one stream each, as a caller of the *_dev API would issue three
        # independent sub-batches -- the library gives each stream its own work scratch, so one operation's last, partly
        # empty wave of workgroups fills up with the next operation's instead of draining alone
        streams = [torch.cuda.Stream(dev) for _ in passes] if len(passes) > 1 and not args.one_stream else None
        cur = torch.cuda.current_stream(dev)

        def step(evs=None, origin=None):
            if origin is not None:
                origin.record()                  # on the current stream, ahead of every pass of the step
                if streams:
                    for st in streams:
                        st.wait_event(origin)
            for j, p in enumerate(passes):
                with torch.cuda.stream(streams[j]) if streams else contextlib.nullcontext():
                    dst = ogs[j].next_buffer() if ogs[j] else outs[j]
                    if evs:
                        evs[j][0].record()
                    p["launch"](dst)
                    if evs:
                        evs[j][1].record()
                    if ogs[j]:
                        ogs[j].submit()          # async RCCL gather of this batch, overlapped with what follows
                        outs[j] = dst

        def finish():
            for og in ogs:
                if og:
                    og.finish()
            if streams:                          # the current stream is behind every pass's stream
                for st in streams:
                    cur.wait_stream(st)

        def ramp_step():                         # kernels only, no gathers
            for j, p in enumerate(passes):
                with torch.cuda.stream(streams[j]) if streams else contextlib.nullcontext():
                    p["launch"](outs[j])

        for _ in range(args.warmup):
            step()
        finish()
        torch.cuda.synchronize()
        # Clock ramp: after an idle gap (the host-side set-up between two blocks, a barrier that waits for a late rank) the
        # chip needs 20-40 ms of load to reach its sustained clock; the first launches run 10-19 % slower
        # (profiles/r04_warmup_probe.txt).  So every rank issues ~CLOCK_RAMP_S of untimed launches of the same pass BEHIND
        # the opening barrier, directly in front of its timed steps: whatever a rank did before the barrier -- rank 0's extra
        # bookkeeping, a slow first-touch -- its timed steps start on a chip that has just been busy for 60 ms.
        t_one = time.perf_counter()
        ramp_step()
        torch.cuda.synchronize()
        one = max(time.perf_counter() - t_one, 1e-4)
        n_ramp = max(1, min(200, int(CLOCK_RAMP_S / one))) if ramp else 0
        events = [[(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in passes]
                  for _ in range(args.steps)]
        origins = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
        dev_t0, dev_t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        if delay_rank0_s and rank == 0:          # test knob: a late rank 0 must not change anybody's timed region
            time.sleep(delay_rank0_s)
        barrier()
        wall0 = time.perf_counter()
        for _ in range(n_ramp):
            ramp_step()
        torch.cuda.synchronize()                 # queue empty, clocks up; the first timed launch follows within microseconds
        t0 = time.perf_counter()
        dev_t0.record()
        for k in range(args.steps):
            step(events[k], origins[k])
        finish()                                 # every gather of the timed steps has completed inside the timed region
        dev_t1.record()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        barrier()
        wall = time.perf_counter() - wall0       # opening barrier to closing barrier: ramp + timed steps + rank skew
        local = t1 - t0
        elapsed = local
        kms = [sum(events[k][j][0].elapsed_time(events[k][j][1]) for k in range(args.steps)) / max(1, args.steps)
               for j in range(len(passes))]
        # device-side span of the timed steps over all their streams: the first pass's start to the last pass's end (HIP
        # events against one origin), per step -- with one pass per step this is that pass's mean kernel time plus the gaps
        # between steps; with three streams it is what the overlapping kernels took together
        first = min(origins[0].elapsed_time(events[0][j][0]) for j in range(len(passes)))
        last = max(origins[0].elapsed_time(events[args.steps - 1][j][1]) for j in range(len(passes)))
        span_ms = (last - first) / max(1, args.steps)

        # ---- bit-exactness of what was just timed: the last step's output buffers against the reference's digests ----
        want = expected_digests(rank, world, n)
        exact = []
        for j, p in enumerate(passes):
            key = DIGEST_KEY[p["wl"]]
            exp = (want.get("mixed_thirds", {}) if mixed else want).get(key) if want else None
            ok = None if exp is None else bool(digest_of(outs[j]) == exp)
            if ok and p["wl"] == "x25519" and not mixed:             # ... and the clamped secret keys written back in place
                ok = bool(digest_of(p["sk"]) == want["x25519_sk_clamped"])
            exact.append(ok)
        gathered_exact = None
        if use_dist and rank == 0 and not share_gpu:                 # the rows RCCL delivered to the root, rank by rank
            gathered_exact = []
            for j, p in enumerate(passes):
                b = (ogs[j].step - 1) & 1
                per = []
                for r in range(world):
                    w = expected_digests(r, world, n)
                    exp = (w.get("mixed_thirds", {}) if mixed else w).get(DIGEST_KEY[p["wl"]]) if w else None
                    per.append(None if exp is None else bool(digest_of(ogs[j].recv[b][r]) == exp))
                gathered_exact.append(per)

        attribution = None
        if use_dist:
            t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if share_gpu else dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
            # what a sub-linear N-GPU result would be made of: every rank's own kernel time and step time, and the
            # gathers timed ALONE afterwards (K steps of gathers with nothing to hide under; in the timed region above
            # they ran underneath the next batch's kernels)
            barrier()
            g0 = time.perf_counter()
            for _ in range(args.steps):
                for og in ogs:
                    og.next_buffer()
                    og.submit()
            finish()
            barrier()
            gather_ms = (time.perf_counter() - g0) / max(1, args.steps) * 1e3
            mine = {"rank": rank, "kernel_ms": [round(k, 4) for k in kms], "step_ms": round(local / max(1, args.steps) * 1e3, 4),
                    "step_ms_device_events": round(dev_t0.elapsed_time(dev_t1) / max(1, args.steps), 4),
                    "t0_offset_ms": t0, "ramp_launches": n_ramp, "gather_ms_alone": round(gather_ms, 4), "bit_exact": exact}
            everyone = [None] * world
            dist.all_gather_object(everyone, mine)
            base = min(a["t0_offset_ms"] for a in everyone)          # CLOCK_MONOTONIC is shared by the processes of a node
            for a in everyone:
                a["t0_offset_ms"] = round((a["t0_offset_ms"] - base) * 1e3, 3)
            attribution = everyone
        return {"elapsed": elapsed, "kms": kms, "outs": outs, "attribution": attribution, "span_ms": span_ms,
                "bit_exact": exact, "gathered_bit_exact": gathered_exact,
                "timing": {"step_ms_wall_this_rank": round(local / max(1, args.steps) * 1e3, 4),
                           "step_ms_device_events_this_rank": round(dev_t0.elapsed_time(dev_t1) / max(1, args.steps), 4),
                           "ramp_launches": n_ramp, "barrier_to_barrier_ms": round(wall * 1e3, 3)}}

    def summarize(p, run, j=0):
        elapsed, kernel_ms, out, attribution = run["elapsed"], run["kms"][j], run["outs"][j], run["attribution"]
        r = {"value": round(world * p["n"] * args.steps / elapsed, 1), "unit": "ops/s",
             "ms_per_step": round(elapsed / args.steps * 1e3, 4), "steps": args.steps, "warmup": args.warmup,
             "n_gpus": world, "batch_per_gpu": p["n"], "workload": WORKLOAD_NAME[p["wl"]],
             "roofline": roofline_for(p["wl"], p["n"], kernel_ms), "timing": run["timing"]}
        # bit-exactness against the reference's digests: this rank's buffer, or every rank's (N > 1)
        r["bit_exact"] = [a["bit_exact"][j] for a in attribution] if attribution else [run["bit_exact"][j]]
        if run["gathered_bit_exact"]:
            r["gathered_rows_bit_exact"] = run["gathered_bit_exact"][j]
        if attribution:
            r["per_rank"] = [{"rank": a["rank"], "kernel_ms": a["kernel_ms"][j], "step_ms": a["step_ms"],
                              "step_ms_device_events": a["step_ms_device_events"], "t0_offset_ms": a["t0_offset_ms"],
                              "ramp_launches": a["ramp_launches"], "gather_ms_alone": a["gather_ms_alone"],
                              "bit_exact": a["bit_exact"][j]} for a in attribution]
        if p["wl"] == "verify":
            rejected = (out.view(-1) == 0).cpu().numpy()
            r["rejected"] = int(rejected.sum())
            r["rejects_exactly_the_corrupted"] = bool((rejected == p["bad"]).all())
        return r

    wl = args.workload
    side = {}
    runs = []                                              # (name, pass, run) of every timed block, for the bit_exact summary
    if wl == "mixed":
        (x0, x1), (s0, s1), (v0, v1) = mixed_thirds(n)
        passes = [make_x25519(x1 - x0, x0), make_sign(s1 - s0, s0), make_verify(v1 - v0, v0)]
        run = run_timed(passes, mixed=True)
        elapsed, kms = run["elapsed"], run["kms"]
        parts = [summarize(p, run, j) for j, p in enumerate(passes)]
        runs += [(p["wl"], r) for p, r in zip(passes, parts)]
        # on three streams the passes' event spans overlap: the device-side span of the step (earliest start event to latest
        # end event over the three streams) is what the kernels took together -- not the wall time, which also holds the
        # gather submissions and the host's launch overhead
        kernel_ms = sum(kms) if args.one_stream else run["span_ms"]
        bytes_per_launch = sum(BYTES_PER_OP[p["wl"]] * p["n"] for p in passes)
        macs = sum(MACS_PER_OP[p["wl"]] * p["n"] for p in passes)
        peak_mac, peak_src = measured_mad_peak()
        gbs = bytes_per_launch / (kernel_ms * 1e-3) / 1e9
        vfrac = round(macs / (kernel_ms * 1e-3) / peak_mac, 4) if peak_mac else None
        roof = {"bound": "hbm", "achieved": round(gbs, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(gbs / HBM_PEAK_GBS, 6), "traffic": None,
                "kernel": "X25519 pass + sign pass + verify pass (see parts)", "kernel_ms": round(kernel_ms, 4),
                "algorithmic_bytes_per_launch": bytes_per_launch, "binding": "valu", "binding_frac": vfrac,
                "valu": {"bound": "valu v_mad_u64_u32 issue", "achieved": round(macs / (kernel_ms * 1e-3) / 1e12, 4),
                         "peak": round(peak_mac / 1e12, 4) if peak_mac else None, "unit": "T 32x32 MAC/s", "frac": vfrac},
                "streams": 1 if args.one_stream else len(passes),
                "kernel_ms_is": "sum of the passes' event times" if args.one_stream else "device-side span over the streams (events)",
                "parts": {p["wl"]: {"n": p["n"], ("kernel_ms" if args.one_stream else "span_ms_overlapping"): round(k, 4)}
                          for p, k in zip(passes, kms)}}
        primary = {"value": round(world * n * args.steps / elapsed, 1), "ms_per_step": round(elapsed / args.steps * 1e3, 4),
                   "roofline": roof, "timing": run["timing"]}
        side["verify_rejects_exactly_the_corrupted"] = parts[2]["rejects_exactly_the_corrupted"]
    else:
        make = {"x25519": make_x25519, "sign": make_sign, "verify": make_verify}[wl]
        p = make(n)
        primary = summarize(p, run_timed([p]))
        runs.append((wl, primary))
        if wl == "x25519" and not args.no_side:
            # the second half of BASELINE.json's metric, and config 3, with the same protocol
            for name, mk in (("verify", make_verify), ("sign", make_sign)):
                side[name] = summarize(mk(n), run_timed([mk(n)]))
                runs.append((name, side[name]))

    # ---- behind ALL timed blocks: the in-kernel probe.  Every rank measures the shader clock its X25519 ladder runs at
    # (s_memtime / s_memrealtime inside the kernel, no host clock involved), so that a slow rank of an N-GPU run is
    # attributable; rank 0's stamps also give the issue-model fraction of the ladder step.
    probe, clocks = None, None
    if wl in ("x25519", "mixed") and not args.no_side:
        probe = issue_model(n, live=True)
        mine = {"rank": rank, "shader_clock_GHz": (probe or {}).get("shader_clock_GHz"),
                "live": bool(probe and str(probe.get("source", "")).startswith("live"))}
        clocks = [mine]
        if use_dist:
            clocks = [None] * world
            dist.all_gather_object(clocks, mine)
    elif wl == "x25519":
        probe = issue_model(n, live=False)
    if wl == "x25519" and probe:
        attach_probe(primary["roofline"]["valu"], probe)
        primary["roofline"]["binding_frac"] = binding_frac(primary["roofline"]["valu"])

    # ---- ... and the roof itself: the multiplier's issue peak on THIS device in THIS run (rank 0; the other ranks wait at the
    # closing barrier of the last block)
    live_peak = live_mad_peak() if rank == 0 and not args.no_side else None
    if rank == 0:
        for q in [primary] + [side[k] for k in ("verify", "sign") if isinstance(side.get(k), dict)]:
            v = q["roofline"]["valu"]
            apply_live_peak(v, live_peak)
            if wl != "mixed":
                q["roofline"]["binding_frac"] = binding_frac(v)
            else:
                q["roofline"]["binding_frac"] = v.get("frac")

    result = None
    if rank == 0:
        roof = primary["roofline"]
        for name in ("verify", "sign"):
            if name in side and isinstance(side[name], dict):
                q = side[name]
                roof[name] = {"value": q["value"], "unit": "ops/s", "ms_per_step": q["ms_per_step"],
                              "kernel_ms": q["roofline"]["kernel_ms"], "hbm_frac": q["roofline"]["frac"],
                              "achieved_GBps": q["roofline"]["achieved"], "traffic": q["roofline"]["traffic"],
                              "traffic_over_algorithmic": q["roofline"]["traffic_over_algorithmic"],
                              "algorithmic_bytes_per_launch": q["roofline"]["algorithmic_bytes_per_launch"],
                              # two fractions of the MAD roof: the reference's operation count / time (a speed-up where the
                              # device does less work than the reference) and what the kernels really issue / time
                              "binding": "valu", "binding_frac": q["roofline"]["binding_frac"],
                              "valu_frac_algorithmic": q["roofline"]["valu"]["frac"],
                              "valu_frac_executed": q["roofline"]["valu"]["frac_executed"],
                              "executed_macs_per_op": q["roofline"]["valu"]["executed_macs_per_op"],
                              "valu_peak": q["roofline"]["valu"]["peak"], "valu_peak_live": q["roofline"]["valu"]["peak_live"],
                              "valu_busy": (q["roofline"]["valu"].get("issue") or {}).get("valu_busy"),
                              "valu_issue_util": (q["roofline"]["valu"].get("issue") or {}).get("valu_issue_util"),
                              "valu_issue": q["roofline"]["valu"].get("issue"),
                              "kernel": q["roofline"]["kernel"]}
        # bit-exactness of everything this run timed, against the reference's portable-C outputs (committed digests)
        flat = [b for _, r in runs for b in r["bit_exact"]] + [b for _, r in runs for b in r.get("gathered_rows_bit_exact", [])]
        bit_exact = {name: (r["bit_exact"] if world > 1 else r["bit_exact"][0]) for name, r in runs}
        bit_exact["all"] = None if any(b is None for b in flat) else bool(all(flat))
        bit_exact["mismatch"] = any(b is False for b in flat)
        if any("gathered_rows_bit_exact" in r for _, r in runs):
            bit_exact["gathered_rows_at_root"] = {name: r["gathered_rows_bit_exact"] for name, r in runs if "gathered_rows_bit_exact" in r}
        bit_exact["against"] = ("tests/golden/digests.json: SHA-256 of the reference's portable-C outputs (oracle/_ref, written by "
                                "tests/golden/gen_golden.py) for each rank's seeded inputs; the last timed step's output buffers "
                                "(X25519: shared keys and the clamped secret keys)" + ("" if bit_exact["all"] is not None else
                                "; null = no digest committed for this rank / batch size"))
        result = {
            "metric": METRIC_NAME[wl],
            "value": primary["value"], "unit": "ops/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": primary["ms_per_step"], "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u64", "dtype_note": "26/25-bit limbs in u32 registers, 32x32+64->64-bit "
            "integer MACs (v_mad_u64_u32), bit-exact results", "data": "synthetic",
            "bit_exact": bit_exact,
            "clock_ramp_ms": round(CLOCK_RAMP_S * 1e3), "timing_protocol": "per block: W warm-up steps; barrier; ~60 ms of untimed "
            "launches of the same pass (after an idle gap the chip runs 10-19 % slower for its first 20-40 ms); synchronize; t0; K "
            "steps + their gathers; synchronize; t1; barrier.  value = units of all ranks / max over ranks of (t1 - t0); device-event "
            "time between the same points in `timing` / per_rank",
            "config": {"workload": WORKLOAD_NAME[wl], "batch_per_gpu": n, "global_batch": n * world,
                       "parallelism": f"shard{world}" + (("+gloo_gather_SHARED_GPU_SELFTEST" if share_gpu else "+rccl_gather")
                                                          if use_dist else "") +
                                      ("+COMPUTE_PARTITIONS_OF_ONE_GPU_not_a_scaling_figure" if rccl["compute_partitions_of_one_device"] else "")},
            "roofline": roof, "timing": primary.get("timing"),
        }
        result["ranks"] = identities                 # every rank's device (UUID, PCI address), backend and RCCL version
        result["rccl"] = rccl
        for k in ("rejected", "rejects_exactly_the_corrupted"):        # --workload verify: the primary pass's own check
            if k in primary:
                result[k] = primary[k]
        if primary.get("per_rank"):
            result["per_rank"] = primary["per_rank"]
        if wl == "mixed" and parts[0].get("per_rank"):
            result["per_rank"] = {q["wl"]: r.get("per_rank") for q, r in zip(passes, parts)}
        if clocks:
            result["per_rank_shader_clock_GHz"] = [c["shader_clock_GHz"] for c in sorted(clocks, key=lambda c: c["rank"])]
        result.update(side)

    # ---- extra measurements outside the protocol above (rank 0, single GPU, default workload only) ----
    if rank == 0 and world == 1 and not args.no_side and wl == "x25519":
        extra = {}
        e = ed_material()

        def timeit(fn, reps=5):
            fn(); torch.cuda.synchronize()
            w0 = time.perf_counter()                  # the clock ramp of run_timed, for the same reason
            while time.perf_counter() - w0 < CLOCK_RAMP_S:
                fn()
                torch.cuda.synchronize()
            tot = 0.0
            for _ in range(reps):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(); fn(); b.record(); torch.cuda.synchronize()
                tot += a.elapsed_time(b)
            return tot / reps

        pub2, priv2 = torch.empty_like(e["pub"]), torch.empty_like(e["priv"])
        ms = timeit(lambda: eng.api.ed25519_CreateKeyPair_dev(pub2, priv2, e["esk"]))
        extra["ed25519_keypair_per_s"] = round(n / (ms * 1e-3), 1)
        extra["ed25519_keypair_kernel_ms"] = round(ms, 4)
        want = expected_digests(0, 1, n)
        extra["ed25519_keypair_bit_exact"] = (None if not want else
                                              bool(digest_of(pub2) == want["ed25519_pub"] and digest_of(priv2) == want["ed25519_priv"]))
        # two-phase verification, ONE key for the whole batch (Verify_Init once, 2^20 Verify_Check)
        from curve25519_amd import _lib
        import ctypes as C
        L = _lib.load()
        ok = torch.empty((n, 1), dtype=torch.int32, device=dev)
        one_priv = e["priv"][:1].repeat(n, 1).contiguous()
        one_sig = eng.ed25519_sign(one_priv, e["msg"])
        ctx = torch.empty((1, 2080), dtype=torch.uint8, device=dev)
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        _lib.check(L.ed25519_Verify_Init_dev(C.c_void_p(ctx.data_ptr()), C.c_void_p(e["pub"][:1].contiguous().data_ptr()), 1, st),
                   "ed25519_Verify_Init_dev")
        chk = lambda: _lib.check(L.ed25519_Verify_Check_dev(C.c_void_p(ok.data_ptr()), C.c_void_p(ctx.data_ptr()),  # noqa: E731
                                                            C.c_void_p(one_sig.data_ptr()), C.c_void_p(e["msg"].data_ptr()),
                                                            32, n, st), "ed25519_Verify_Check_dev")
        ms = timeit(chk)
        extra["ed25519_verify_check_one_key_per_s"] = round(n / (ms * 1e-3), 1)
        extra["ed25519_verify_check_one_key_all_valid"] = bool(int(ok.sum().item()) == n)
        # (a big batch under one on-curve key walks two wide combs, the key's built per call: engine.hip,
        # k_ed25519_verify_check_wide; the reference's own operation order for the same batch, tunable ONE_KEY_WIDE = 0:)
        with _lib.tunable("ONE_KEY_WIDE", 0):
            ms = timeit(chk)
        extra["ed25519_verify_check_one_key_reference_order_per_s"] = round(n / (ms * 1e-3), 1)
        extra["ed25519_verify_check_one_key_reference_order_all_valid"] = bool(int(ok.sum().item()) == n)
        # BASELINE.json configs[3] AS WORDED ("double-scalar, 4-fold"): every element through the reference's own operation
        # order -- Verify_Init's 16-row 4-fold table per key, then the 4-fold + 8-fold walk and one shared inversion
        # (ed25519_verify.c:179-313) -- instead of the shipped lattice-shortened walk; same verdicts, public data
        vq = make_verify(n)
        with _lib.tunable("VERIFY_REFERENCE_ORDER", 1):
            vr = run_timed([vq])
        rej = (vr["outs"][0].view(-1) == 0).cpu().numpy()
        extra["verify_reference_order_per_s"] = round(n * args.steps / vr["elapsed"], 1)
        extra["verify_reference_order_kernel_ms"] = round(vr["kms"][0], 4)
        extra["verify_reference_order_bit_exact"] = vr["bit_exact"][0]
        extra["verify_reference_order_rejects_exactly_the_corrupted"] = bool((rej == vq["bad"]).all())
        # BASELINE.json configs[2] AS WORDED ("8-fold base_folding8 table staged in LDS"): the 8 x 32 signed comb, eight tables in
        # 120 KiB of LDS per workgroup, instead of the shipped wide comb read through L2 (tunable BASE_COMB); same bytes
        sq = make_sign(n)
        with _lib.tunable("BASE_COMB", 0):
            sr = run_timed([sq])
        extra["sign_lds_comb_per_s"] = round(n * args.steps / sr["elapsed"], 1)
        extra["sign_lds_comb_kernel_ms"] = round(sr["kms"][0], 4)
        extra["sign_lds_comb_bit_exact"] = sr["bit_exact"][0]
        # the same sign / verify blocks WITHOUT the clock ramp (the round-3 protocol): what part of a round-over-round change
        # is the protocol's and what part the kernels'
        off = {}
        for name, mk in (("verify", make_verify), ("sign", make_sign)):
            time.sleep(0.2)                                   # the idle gap the ramp exists for
            r0 = run_timed([mk(n)], ramp=False)
            off[name + "_per_s"] = round(n * args.steps / r0["elapsed"], 1)
        extra["clock_ramp_off"] = off
        # calls of a few thousand elements (four lanes per element: csrc/quad25519.cuh) and ONE call through the reference's own
        # prototypes (one operation per wave, the completion word): what a caller that does not gather 2^17 operations per call gets.
        # Device-resident prefixes of the bench's own arrays, HIP events; single calls: wall clock around the C call, host pointers.
        mid = {}
        xs = {"sk": up(synth.random_bytes((1 << 14, 32), synth.SEED_X25519_SK + seed_shift)),
              "pk": up(synth.random_bytes((1 << 14, 32), synth.SEED_X25519_PK + seed_shift))}
        for lg in (12, 14):
            m = 1 << lg
            out_m = torch.empty((m, 32), dtype=torch.uint8, device=dev)
            sig_m, ok_m = torch.empty((m, 64), dtype=torch.uint8, device=dev), torch.empty((m, 1), dtype=torch.int32, device=dev)
            skc = xs["sk"][:m].clone()
            calls = {"x25519": lambda: eng.api.curve25519_dh_CreateSharedKey_dev(out_m, xs["pk"][:m], skc),
                     "sign": lambda: eng.api.ed25519_SignMessage_dev(sig_m, e["priv"][:m], e["msg"][:m]),
                     "verify": lambda: eng.api.ed25519_VerifySignature_dev(ok_m, e["sig"][:m], e["pub"][:m], e["msg"][:m])}
            for name, fn in calls.items():
                mid.setdefault(name, {})[f"2^{lg}"] = {"ms_per_call": round(timeit(fn, reps=20), 4)}
                mid[name][f"2^{lg}"]["per_s"] = round(m / (mid[name][f"2^{lg}"]["ms_per_call"] * 1e-3), 1)
            mid["verify"][f"2^{lg}"]["all_valid"] = bool(int(ok_m.sum().item()) == m)
        extra["mid_size_calls"] = mid
        extra["single_call_us"] = single_call_latencies(L, e)
        result["extra"] = extra

    if use_dist:
        dist.destroy_process_group()
    if rank == 0:
        result["cpu_baseline"] = None if args.no_cpu else cpu_baseline(quick=world > 1)
        os.write(result_fd, (json.dumps(result) + "\n").encode())
        if result["bit_exact"]["mismatch"]:
            print("bench.py: OUTPUTS DIFFER from the reference's committed digests: " + json.dumps(result["bit_exact"]),
                  file=sys.stderr)
            sys.exit(3)


if __name__ == "__main__":
    main()
