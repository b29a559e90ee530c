#!/usr/bin/env python3
"""tools/small_batch_sweep.py -- latency of a call of n device-resident elements, one operation per wave (coop25519.cuh)
against one operation per lane (the batch kernels' narrow shapes), around the crossover the tunable COOP_MAX sets."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from curve25519_amd import _lib, api, synth  # noqa: E402

dev = torch.device("cuda", 0)
N = 1 << 14
up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
sk, pk = (up(a) for a in synth.x25519_inputs(N))
esk, msg = synth.ed25519_inputs(N)
pub, priv = api.ed25519_CreateKeyPair(esk)
desk, dpriv, dmsg = up(esk), up(priv), up(msg)
o32, o64, p32, p64 = (torch.empty((N, w), dtype=torch.uint8, device=dev) for w in (32, 64, 32, 64))
dsig, dpub = up(api.ed25519_SignMessage(priv, msg)), up(pub)
ok = torch.empty((N, 1), dtype=torch.int32, device=dev)
ops = {
    "x25519": lambda n: api.curve25519_dh_CreateSharedKey_dev(o32[:n], pk[:n], sk[:n]),
    "public_fast": lambda n: api.curve25519_dh_CalculatePublicKey_dev(o32[:n], sk[:n], fast=True),
    "keypair": lambda n: api.ed25519_CreateKeyPair_dev(p32[:n], p64[:n], desk[:n]),
    "sign": lambda n: api.ed25519_SignMessage_dev(o64[:n], dpriv[:n], dmsg[:n]),
    "verify": lambda n: api.ed25519_VerifySignature_dev(ok[:n], dsig[:n], dpub[:n], dmsg[:n]),
}


def us(fn, n, reps=12):
    fn(n); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(n); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    return sorted(ts)[len(ts) // 2]


print(f"{'op':12s} {'n':>6s} {'two waves each [us]':>20s} {'one per wave [us]':>18s}     (curve25519_dh_CreateSharedKey: tunable LADDER2_MAX)")
for n in (1, 16, 64, 128, 256, 512, 1024, 2048):
    _lib.set_tunable("COOP_MAX", 1 << 20)
    _lib.set_tunable("LADDER2_MAX", 1 << 20)
    a = us(ops["x25519"], n)
    _lib.set_tunable("LADDER2_MAX", 0)
    b = us(ops["x25519"], n)
    print(f"{'x25519':12s} {n:6d} {a:20.1f} {b:18.1f}", flush=True)
_lib.set_tunable("LADDER2_MAX", -1)
print(f"{'op':12s} {'n':>6s} {'one per wave [us]':>18s} {'one per lane [us]':>18s}")
for name, fn in ops.items():
    for n in (1, 16, 64, 256, 1024, 2048, 4096, 8192, 16384):
        _lib.set_tunable("COOP_MAX", 1 << 20)
        c = us(fn, n)
        _lib.set_tunable("COOP_MAX", 0)
        b = us(fn, n)
        print(f"{name:12s} {n:6d} {c:18.1f} {b:18.1f}", flush=True)
_lib.set_tunable("COOP_MAX", -1)
