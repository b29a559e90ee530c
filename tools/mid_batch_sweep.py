#!/usr/bin/env python3
"""tools/mid_batch_sweep.py -- throughput of one *_dev call of n device-resident elements for n = 2^10 .. 2^17: the sizes between
"one operation per wave" (coop25519.cuh) and "the chip is full of one-lane-per-element waves" (2^16 elements = one wave per
SIMD), where the quad kernels (quad25519.cuh: four lanes per element) run.

    python tools/mid_batch_sweep.py [--ops x25519,sign,keypair,verify] [--exps 10,11,...,17] [COLUMN=KNOB:VAL,KNOB:VAL ...]

Every COLUMN is the shipped library under a set of tunables (include/curve25519_amd.h: c25519_amd_tunable_set); the first,
implicit column "shipped" has none.  E.g.  lane=QUAD_MAX:0  for the one-lane-per-element / one-per-wave kernels only.
Timing: per (n, op, column) the call is repeated for ~40 ms untimed (sustained clock), then a burst between two HIP events;
min over rounds.  Outputs are compared between columns (identical bytes) before anything is printed."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from curve25519_amd import _lib, api, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("columns", nargs="*")
ap.add_argument("--ops", default="x25519,sign,keypair,verify")
ap.add_argument("--exps", default="10,11,12,13,14,15,16,17")
ap.add_argument("--rounds", type=int, default=3)
args = ap.parse_args()

cols = [("shipped", {})]
for c in args.columns:
    name, _, kv = c.partition("=")
    cols.append((name, {k: int(v) for k, v in (p.split(":") for p in kv.split(",") if p)}))
ALL_KNOBS = sorted({k for _, kn in cols for k in kn})

dev = torch.device("cuda", 0)
exps = [int(e) for e in args.exps.split(",")]
N = 1 << max(exps)
up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
sk, pk = (up(a) for a in synth.x25519_inputs(N))
esk, msg = synth.ed25519_inputs(N)
pub, priv = api.ed25519_CreateKeyPair(esk)
desk, dpriv, dmsg = up(esk), up(priv), up(msg)
o32, o64, p32, p64 = (torch.empty((N, w), dtype=torch.uint8, device=dev) for w in (32, 64, 32, 64))
sig_np = api.ed25519_SignMessage(priv, msg)
bsig, bmsg, bad = synth.corrupt_for_verify(sig_np, msg)
dsig, dpub, dbmsg = up(bsig), up(pub), up(bmsg)
ok = torch.empty((N, 1), dtype=torch.int32, device=dev)
ops = {
    "x25519": (lambda n: api.curve25519_dh_CreateSharedKey_dev(o32[:n], pk[:n], sk[:n]), lambda n: o32[:n]),
    "public": (lambda n: api.curve25519_dh_CalculatePublicKey_dev(o32[:n], sk[:n]), lambda n: o32[:n]),
    "public_fast": (lambda n: api.curve25519_dh_CalculatePublicKey_dev(o32[:n], sk[:n], fast=True), lambda n: o32[:n]),
    "keypair": (lambda n: api.ed25519_CreateKeyPair_dev(p32[:n], p64[:n], desk[:n]), lambda n: p64[:n]),
    "sign": (lambda n: api.ed25519_SignMessage_dev(o64[:n], dpriv[:n], dmsg[:n]), lambda n: o64[:n]),
    "verify": (lambda n: api.ed25519_VerifySignature_dev(ok[:n], dsig[:n], dpub[:n], dbmsg[:n]), lambda n: ok[:n]),
}


def set_knobs(kn):
    for k in ALL_KNOBS:
        _lib.set_tunable(k, -1)
    for k, v in kn.items():
        _lib.set_tunable(k, v)


def ms_of(fn, n):
    fn(n); torch.cuda.synchronize()
    w0 = time.perf_counter()
    while time.perf_counter() - w0 < 0.04:
        fn(n)
        torch.cuda.synchronize()
    burst = 8 if n <= (1 << 14) else 4
    best = None
    for _ in range(args.rounds):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(burst):
            fn(n)
        b.record(); torch.cuda.synchronize()
        t = a.elapsed_time(b) / burst
        best = t if best is None else min(best, t)
    return best


print("# tools/mid_batch_sweep.py on", torch.cuda.get_device_name(0), "-- ms per call | M ops/s, per column; columns:",
      "; ".join(f"{n} {kn}" for n, kn in cols))
print(f"{'op':12s} {'n':>7s} " + " ".join(f"{n:>22s}" for n, _ in cols))
for name in args.ops.split(","):
    fn, res = ops[name]
    for e in exps:
        n = 1 << e
        cells, ref = [], None
        for cname, kn in cols:
            set_knobs(kn)
            t = ms_of(fn, n)
            h = hash(res(n).cpu().numpy().tobytes())
            if ref is None:
                ref = h
            assert h == ref, f"{name} n=2^{e}: column {cname} produced different bytes"
            cells.append(f"{t:9.3f} ms {n / t / 1e3:8.1f} M/s")
        print(f"{name:12s} 2^{e:<5d} " + " ".join(f"{c:>22s}" for c in cells), flush=True)
set_knobs({})
