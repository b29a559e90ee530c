#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path over one batch: 2^20 curve25519_dh_CreateSharedKey operations per
GPU (BASELINE.json configs[1]), inputs already resident in HBM, followed for N > 1 by the single RCCL
gather of the 32-byte results to rank 0 that north_star names.  Weak scaling: every rank owns its own
2^20 keypairs.  Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  roofline      -- the X25519 kernel against the HBM roof the contract asks for (frac << 1 by
                   construction: the path is VALU-integer bound) ...
  roofline_valu -- ... and against the measured v_mad_u64_u32 issue peak (profiles/r01_valu_rates.json),
                   which is the roof that actually binds;
  cpu_baseline  -- the reference's portable-C path (oracle/_ref) or the oracle port timed on this host;
  extra         -- Ed25519 sign / verify throughput at the same batch size (configs[2], configs[3]).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BATCH = 1 << 20
BYTES_PER_OP = {"x25519": 96, "sign": 160, "verify": 132}          # SURVEY.md 8(d), compulsory HBM bytes
MACS_PER_OP = {"x25519": 184104, "sign": 52992, "verify": 245664}  # SURVEY.md 8(a), 32x32 MACs at 72/mul
HBM_PEAK_GBS = 8000.0                                               # MI355X_MICROARCH.md


def measured_mad_peak():
    """lane-MAC/s of v_mad_u64_u32 measured by tools/ubench/valu_rates on MI355X (committed summary)."""
    try:
        with open(os.path.join(ROOT, "profiles", "r01_valu_rates.json")) as f:
            rates = json.load(f)["rates"]
        return max(v for k, v in rates.items() if k.startswith("v_mad_u64_u32"))
    except Exception:
        return None


PASS_KERNELS = {
    "x25519": ("k_x25519_fused",),
    "sign": ("k_ed25519_sign_mult", "k_batch_invert<FinishPack>", "k_ed25519_sign_finish"),
    "verify": ("k_ed25519_verify_init<c25519::QTableLimbs>", "k_ed25519_verify_check<c25519::QTableLimbs>",
               "k_batch_invert<FinishVerify>"),
}
METRIC_NAME = {
    "x25519": "X25519 shared-key ops/sec (batch=2^20 per GPU, variable-base Montgomery ladder)",
    "sign": "Ed25519 signs/sec (batch=2^20 per GPU, 8-fold fixed-base walk, 32-byte messages)",
    "verify": "Ed25519 verifies/sec (batch=2^20 per GPU, distinct keys, 4-fold + 8-fold double-scalar walk)",
}
WORKLOAD_NAME = {
    "x25519": "BASELINE.json configs[1]: batch 2^20 X25519 curve25519_dh_CreateSharedKey per GPU, one keypair per "
              "lane, inputs resident in HBM",
    "sign": "BASELINE.json configs[2]: batch 2^20 ed25519_SignMessage per GPU, base table staged in LDS",
    "verify": "BASELINE.json configs[3]: batch 2^20 ed25519_VerifySignature per GPU (Verify_Init + Verify_Check)",
}


def measured_traffic(kernels):
    """HBM bytes per X25519 pass (ladder launch + batched-inversion launch) from the committed rocprofv3 PMC
    passes (separate --pmc runs of this same bench, summarised by tools/rocpd_summary.py):
    WRITE_SIZE + 2 x FETCH_SIZE, both in KiB -- the x2 is the gfx950 FETCH_SIZE correction of
    MI355X_MICROARCH.md (HBM section)."""
    try:
        import glob
        path = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc.json")))[-1]
        with open(path) as f:
            d = json.load(f)
        def rec(name):                      # rocprof prints template kernels as "void name<...>"
            hits = [v for k, v in d.items() if name in k and "<true>" not in k]
            return hits[0]
        total = sum((2.0 * rec(k)["FETCH_SIZE"] + rec(k)["WRITE_SIZE"]) * 1024 for k in kernels)
        return int(total), os.path.basename(path)
    except Exception:
        return None, None


def cpu_baseline(n_per_thread=8192):
    """Time curve25519_dh_CreateSharedKey (plus Ed25519 sign / verify) on the host cores: the real reference
    (portable-C build, oracle/_ref) when its prebuilt library travelled with the repo, else the oracle port.
    A C thread pool fans one contiguous slice per core; bounded sample, same input distribution as the GPU
    workload."""
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
    n = n_per_thread * cores
    sk, pk = synth.x25519_inputs(n)
    esk, msg = synth.ed25519_inputs(cores * 256)
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

    def rate(fn, count):
        t0 = time.perf_counter()
        out = fn()
        return count / (time.perf_counter() - t0), out

    single, _ = rate(lambda: shared(pk[:4096], sk[:4096], 1), 4096)         # config 1: 4096 sequential calls
    # all cores, in rounds of 1024 per thread, until the sample is used up or ~10 s have gone by
    done, t0, chunk = 0, time.perf_counter(), 1024 * cores
    while done < n and time.perf_counter() - t0 < 10.0:
        shared(pk[done:done + chunk], sk[done:done + chunk], cores)
        done += chunk
    multi, n = done / (time.perf_counter() - t0), done
    sign_rate, sig = rate(lambda: sign(cores), cores * 256)
    verify_rate, ok = rate(lambda: verify(sig, cores), cores * 256)
    model = ""
    try:
        with open("/proc/cpuinfo") as f:
            model = next(l.split(":", 1)[1].strip() for l in f if l.startswith("model name"))
    except Exception:
        pass
    return {"value": round(multi, 1), "unit": "X25519 shared-key ops/s", "cores": cores, "kind": kind,
            "sample": f"{n} curve25519_dh_CreateSharedKey calls over {cores} threads "
                      f"(C thread pool, one contiguous slice each); seeded uniform sk/pk",
            "single_core_ops_per_s": round(single, 1), "ed25519_sign_per_s": round(sign_rate, 1),
            "ed25519_verify_per_s": round(verify_rate, 1), "ed25519_verify_all_valid": bool(ok.all()),
            "host_logical_cpus": os.cpu_count(), "cgroup_cpu_quota": quota,
            "cpu_model": model}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=BATCH, help="operations per GPU per step (default 2^20)")
    ap.add_argument("--no-extra", action="store_true", help="skip the Ed25519 sign/verify side measurements")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline")
    ap.add_argument("--workload", choices=("x25519", "sign", "verify"), default="x25519",
                    help="x25519 = BASELINE.json configs[1] (the default and the driver's contract); sign / verify = "
                         "configs[2] / configs[3], same batch size, for the side tables of DESIGN.md")
    ap.add_argument("--dist-selftest", action="store_true",
                    help="run the N>1 code path (process group + RCCL gather) with a world of one rank")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from curve25519_amd import synth
    from curve25519_amd.sharded import HipEngine, OverlappedGather

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: torch.cuda.is_available() is False (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or args.dist_selftest
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    n = args.batch
    eng = HipEngine(dev)
    wl = args.workload
    seed_shift = 0x100 * rank if world > 1 else 0          # every rank owns its own 2^20 elements (weak scaling)
    if wl == "x25519":
        sk = torch.from_numpy(synth.random_bytes((n, 32), synth.SEED_X25519_SK + seed_shift)).to(dev)
        pk = torch.from_numpy(synth.random_bytes((n, 32), synth.SEED_X25519_PK + seed_shift)).to(dev)
        width, odtype = 32, torch.uint8
        launch = lambda dst: eng.api.curve25519_dh_CreateSharedKey_dev(dst, pk, sk)          # noqa: E731
    else:
        esk = torch.from_numpy(synth.random_bytes((n, 32), synth.SEED_ED_SK + seed_shift)).to(dev)
        msg = torch.from_numpy(synth.random_bytes((n, 32), synth.SEED_ED_MSG + seed_shift)).to(dev)
        pub, priv = eng.ed25519_keypair(esk)
        if wl == "sign":
            width, odtype = 64, torch.uint8
            launch = lambda dst: eng.api.ed25519_SignMessage_dev(dst, priv, msg)             # noqa: E731
        else:                                       # config 4: valid signatures + the seeded 1/64 corrupted ones
            sig_np, msg_np, _bad = synth.corrupt_for_verify(eng.ed25519_sign(priv, msg).cpu().numpy(), msg.cpu().numpy())
            sig, msg = torch.from_numpy(sig_np).to(dev), torch.from_numpy(msg_np).to(dev)
            width, odtype = 1, torch.int32
            launch = lambda dst: eng.api.ed25519_VerifySignature_dev(dst, sig, pub, msg)     # noqa: E731
    out = torch.empty((n, width), dtype=odtype, device=dev)

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    og = OverlappedGather(n, width, dev, root=0, dtype=odtype) if use_dist else None

    def step(ev=None):
        dst = og.next_buffer() if og else out
        if ev:
            ev[0].record()
        launch(dst)
        if ev:
            ev[1].record()
        if og:
            og.submit()                  # async RCCL gather of this batch, overlapped with the next batch

    for _ in range(args.warmup):
        step()
    if og:
        og.finish()
    events = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(events[k])
    if og:
        og.finish()                      # every gather of the timed steps has completed inside the timed region
    barrier()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kernel_ms = sum(a.elapsed_time(b) for a, b in events) / max(1, args.steps)

    result = None
    if rank == 0:
        value = world * n * args.steps / elapsed
        kernel_s = kernel_ms * 1e-3
        achieved_gbs = BYTES_PER_OP[wl] * n / kernel_s / 1e9
        peak_mac = measured_mad_peak()
        traffic, traffic_src = measured_traffic(PASS_KERNELS[wl])
        achieved_mac = MACS_PER_OP[wl] * n / kernel_s
        result = {
            "metric": METRIC_NAME[wl],
            "value": round(value, 1), "unit": "ops/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u64", "dtype_note": "26/25-bit limbs in u32 registers, 32x32+64->64-bit "
            "integer MACs (v_mad_u64_u32), bit-exact results", "data": "synthetic",
            "config": {"workload": WORKLOAD_NAME[wl],
                       "batch_per_gpu": n, "global_batch": n * world,
                       "parallelism": f"shard{world}" + ("+rccl_gather" if world > 1 else "")},
            "roofline": {"bound": "hbm", "achieved": round(achieved_gbs, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved_gbs / HBM_PEAK_GBS, 6), "traffic": traffic,
                         "traffic_source": f"profiles/{traffic_src}: (2*FETCH_SIZE + WRITE_SIZE) KiB per pass; for X25519 = "
                                           "96 B/op API bytes + the 32 B/op clamped-key write-back the reference's "
                                           "IN/OUT sk requires" if traffic_src else None,
                         "kernel": " + ".join(PASS_KERNELS[wl]),
                         "kernel_ms": round(kernel_ms, 4),
                         "algorithmic_bytes_per_launch": BYTES_PER_OP[wl] * n,
                         "note": "VALU-integer bound path: HBM fraction is tiny by construction, see roofline_valu"},
            "roofline_valu": {"bound": "valu v_mad_u64_u32", "achieved": round(achieved_mac / 1e12, 4),
                              "peak": round(peak_mac / 1e12, 4) if peak_mac else None, "unit": "T 32x32 MAC/s",
                              "frac": round(achieved_mac / peak_mac, 4) if peak_mac else None,
                              "algorithmic_macs_per_op": MACS_PER_OP[wl]},
        }

    # ---- side measurements, outside the timed region (rank 0, single GPU only) ----
    if rank == 0 and world == 1 and not args.no_extra and wl == "x25519":
        extra = {}
        esk_np, msg_np = synth.ed25519_inputs(n)
        esk = torch.from_numpy(esk_np).to(dev)
        msg = torch.from_numpy(msg_np).to(dev)
        pub, priv = eng.ed25519_keypair(esk)
        sig = eng.ed25519_sign(priv, msg)
        torch.cuda.synchronize()

        def timeit(fn, reps=3):
            fn(); torch.cuda.synchronize()
            best = 1e30
            for _ in range(reps):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(); fn(); b.record(); torch.cuda.synchronize()
                best = min(best, a.elapsed_time(b))
            return best

        # config 4's verify set: the signatures above with the seeded 1/64 sprinkle of corrupted entries
        vsig_np, vmsg_np, bad = synth.corrupt_for_verify(sig.cpu().numpy(), msg_np)
        vsig, vmsg = torch.from_numpy(vsig_np).to(dev), torch.from_numpy(vmsg_np).to(dev)
        ok = torch.empty((n, 1), dtype=torch.int32, device=dev)
        for name, fn in (("sign", lambda: eng.api.ed25519_SignMessage_dev(sig, priv, msg)),
                         ("verify", lambda: eng.api.ed25519_VerifySignature_dev(ok, vsig, pub, vmsg)),
                         ("keypair", lambda: eng.api.ed25519_CreateKeyPair_dev(pub, priv, esk))):
            ms = timeit(fn)
            extra[f"ed25519_{name}_per_s"] = round(n / (ms * 1e-3), 1)
            extra[f"ed25519_{name}_kernel_ms"] = round(ms, 4)
        rejected = (ok.view(-1) == 0).cpu().numpy()
        extra["ed25519_verify_rejected"] = int(rejected.sum())
        extra["ed25519_verify_rejects_exactly_the_corrupted"] = bool((rejected == bad).all())
        # two-phase verification, ONE key for the whole batch (Verify_Init once, 2^20 Verify_Check)
        from curve25519_amd import _lib
        import ctypes as C
        L = _lib.load()
        one_priv = priv[:1].repeat(n, 1).contiguous()
        one_sig = eng.ed25519_sign(one_priv, msg)
        ctx = torch.empty((1, 2080), dtype=torch.uint8, device=dev)
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        _lib.check(L.ed25519_Verify_Init_dev(C.c_void_p(ctx.data_ptr()), C.c_void_p(pub[:1].contiguous().data_ptr()), 1, st),
                   "ed25519_Verify_Init_dev")
        chk = lambda: _lib.check(L.ed25519_Verify_Check_dev(C.c_void_p(ok.data_ptr()), C.c_void_p(ctx.data_ptr()),  # noqa: E731
                                                            C.c_void_p(one_sig.data_ptr()), C.c_void_p(msg.data_ptr()),
                                                            32, n, st), "ed25519_Verify_Check_dev")
        ms = timeit(chk)
        extra["ed25519_verify_check_one_key_per_s"] = round(n / (ms * 1e-3), 1)
        extra["ed25519_verify_check_one_key_all_valid"] = bool(int(ok.sum().item()) == n)
        del vsig, vmsg
        result["extra"] = extra

    if rank == 0 and world == 1 and not args.no_cpu:
        result["cpu_baseline"] = cpu_baseline()
    elif rank == 0:
        result["cpu_baseline"] = None

    if rank == 0:
        print(json.dumps(result))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
