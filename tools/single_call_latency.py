#!/usr/bin/env python3
"""tools/single_call_latency.py -- wall time of the reference's own single-call API on this library (a device batch of one
per call: include/curve25519_dh.h, include/ed25519_signature.h), mean of 200 calls each after 20 warm-up calls."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from curve25519_amd import _lib  # noqa: E402

L = _lib.load()
buf = lambda n, fill=0: (C.c_ubyte * n)(*([fill] * n))  # noqa: E731
sk, pk, shared = buf(32, 7), buf(32, 9), buf(32)
esk, pub, priv, sig, msg = buf(32, 3), buf(32), buf(64), buf(64), buf(32, 5)
L.ed25519_CreateKeyPair(pub, priv, None, esk)
L.ed25519_SignMessage(sig, priv, None, msg, 32)
assert L.ed25519_VerifySignature(sig, pub, msg, 32) == 1
ctx = buf(2080)
L.ed25519_Verify_Init.restype = C.c_void_p
L.ed25519_Verify_Init(ctx, pub)
assert L.ed25519_Verify_Check(ctx, sig, msg, 32) == 1
bctx = buf(192)
L.ed25519_Blinding_Init.restype = C.c_void_p
L.ed25519_Blinding_Init(bctx, msg, 32)
ops = {
    "curve25519_dh_CreateSharedKey": lambda: L.curve25519_dh_CreateSharedKey(shared, pk, sk),
    "curve25519_dh_CalculatePublicKey": lambda: L.curve25519_dh_CalculatePublicKey(shared, sk),
    "ed25519_CreateKeyPair": lambda: L.ed25519_CreateKeyPair(pub, priv, None, esk),
    "ed25519_SignMessage": lambda: L.ed25519_SignMessage(sig, priv, None, msg, 32),
    "ed25519_VerifySignature": lambda: L.ed25519_VerifySignature(sig, pub, msg, 32),
    "curve25519_dh_CalculatePublicKey_fast": lambda: L.curve25519_dh_CalculatePublicKey_fast(shared, sk),
    "ed25519_Verify_Init": lambda: L.ed25519_Verify_Init(ctx, pub),
    "ed25519_Verify_Check": lambda: L.ed25519_Verify_Check(ctx, sig, msg, 32),
    "ed25519_SignMessage (blinded)": lambda: L.ed25519_SignMessage(sig, priv, bctx, msg, 32),
    "ed25519_CreateKeyPair (blinded)": lambda: L.ed25519_CreateKeyPair(pub, priv, bctx, esk),
    "ed25519_Blinding_Init": lambda: L.ed25519_Blinding_Init(bctx, msg, 32),
}
for name, fn in ops.items():
    for _ in range(20):
        fn()
    t = time.perf_counter()
    for _ in range(200):
        fn()
    dt = (time.perf_counter() - t) / 200
    print(f"{name:34s} {dt * 1e6:8.1f} us per call  = {1 / dt:8.0f} calls/s")
