#!/usr/bin/env python3
"""tools/single_call_breakdown.py -- where the time of ONE operation through the reference's prototypes goes: the drop-in
call (host pointers: staging + upload + kernels + download + synchronise), the same kernels on device-resident
buffers (call + synchronise), and the kernels alone (HIP events)."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from curve25519_amd import _lib, api, synth  # noqa: E402

L = _lib.load()
dev = torch.device("cuda", 0)
buf = lambda n, fill=0: (C.c_ubyte * n)(*([fill] * n))  # noqa: E731
sk, pk, shared = buf(32, 7), buf(32, 9), buf(32)
esk, pub, priv, sig, msg = buf(32, 3), buf(32), buf(64), buf(64), buf(32, 5)
L.ed25519_CreateKeyPair(pub, priv, None, esk)
L.ed25519_SignMessage(sig, priv, None, msg, 32)
d = lambda b: torch.from_numpy(np.frombuffer(bytes(b), np.uint8).reshape(1, -1).copy()).to(dev)  # noqa: E731
dsk, dpk, dsh, dpriv, dmsg, dsig, dpub = d(sk), d(pk), d(shared), d(priv), d(msg), d(sig), d(pub)
dok = torch.empty((1, 1), dtype=torch.int32, device=dev)


def wall(fn, reps=200):
    for _ in range(20):
        fn()
    t = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t) / reps * 1e6


def events(fn, reps=50):
    fn(); torch.cuda.synchronize()
    tot = 0.0
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        tot += a.elapsed_time(b)
    return tot / reps * 1e3


def sync(fn):
    def g():
        fn()
        torch.cuda.synchronize()
    return g


rows = (
    ("curve25519_dh_CreateSharedKey", lambda: L.curve25519_dh_CreateSharedKey(shared, pk, sk),
     lambda: api.curve25519_dh_CreateSharedKey_dev(dsh, dpk, dsk)),
    ("ed25519_SignMessage", lambda: L.ed25519_SignMessage(sig, priv, None, msg, 32),
     lambda: api.ed25519_SignMessage_dev(dsig, dpriv, dmsg)),
    ("ed25519_VerifySignature", lambda: L.ed25519_VerifySignature(sig, pub, msg, 32),
     lambda: api.ed25519_VerifySignature_dev(dok, dsig, dpub, dmsg)),
)
print(f"{'one operation':32s} {'drop-in call':>14s} {'_dev + sync':>14s} {'kernels (events)':>18s}   [us]")
for name, host, devfn in rows:
    print(f"{name:32s} {wall(host):14.1f} {wall(sync(devfn)):14.1f} {events(devfn):18.1f}")
noop = lambda: None  # noqa: E731
print(f"{'(empty stream: event pair)':32s} {'':14s} {'':14s} {events(noop):18.1f}")
