#!/usr/bin/env python3
"""tools/multi_virtual_rate.py [--devices 8] [--n 1048576] -- the *_multi entry points over D VIRTUAL devices on ONE GPU (a
device list naming device 0 D times: D workers, shards, pipelines and gather streams, device-to-device copies in RCCL's
place) against the single-GPU *_batch call on the same arrays, alternating; pageable and page-locked caller arrays, both
gather modes.  What it prices is the D > 1 host path's overhead on a box with one GPU, where the D pipelines' 4 D streams
share the process' hardware queues (GPU_MAX_HW_QUEUES, 4 by default: run it again with GPU_MAX_HW_QUEUES=16 in the
environment to give every virtual device queues of its own, as real devices have)."""
import argparse
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from curve25519_amd import _lib, api, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--devices", type=int, default=8)
ap.add_argument("--n", type=int, default=1 << 20)
args = ap.parse_args()
n, D = args.n, args.devices
L = _lib.load()
P = lambda a: C.c_void_p(a.ctypes.data)  # noqa: E731


def pair(fa, fb, reps=9):
    fa(); fb()
    ta, tb = [], []
    for _ in range(reps):
        t = time.perf_counter(); fa(); ta.append(time.perf_counter() - t)
        t = time.perf_counter(); fb(); tb.append(time.perf_counter() - t)
    return sorted(ta)[len(ta) // 2], sorted(tb)[len(tb) // 2]


sk0, pk0 = synth.x25519_inputs(n)
esk, msg0 = synth.ed25519_inputs(n)
pub0, priv0 = api.ed25519_CreateKeyPair(esk)
sig0 = api.ed25519_SignMessage(priv0, msg0)
print(f"# {D} virtual devices on one GPU, n = {n}, GPU_MAX_HW_QUEUES={os.environ.get('GPU_MAX_HW_QUEUES', '(default 4)')}, "
      f"{L.c25519_amd_usable_cpus()} usable CPUs")
for locked in (False, True):
    mk = (lambda a: synth.page_aligned(a.shape, a.dtype, like=a)) if locked else (lambda a: a.copy())
    sk, pk, msg, pub, priv, sig = (mk(a) for a in (sk0, pk0, msg0, pub0, priv0, sig0))
    o32m, o32b = mk(np.zeros((n, 32), np.uint8)), mk(np.zeros((n, 32), np.uint8))
    o64m, o64b = mk(np.zeros((n, 64), np.uint8)), mk(np.zeros((n, 64), np.uint8))
    okm, okb = mk(np.zeros(n, np.int32)), mk(np.zeros(n, np.int32))
    arrays = (sk, pk, msg, pub, priv, sig, o32m, o32b, o64m, o64b, okm, okb)
    if locked:
        for a in arrays:
            assert L.c25519_amd_host_register(P(a), synth.locked_bytes(a)) == 0
    for gather in (1, 0):
        h = C.c_void_p()
        assert L.c25519_amd_multi_create(C.byref(h), (C.c_int * D)(*([0] * D)), D) == 0
        assert L.c25519_amd_multi_set_gather(h, gather) == 0
        for name, fm, fb, a, b in (
                ("x25519", lambda: L.curve25519_dh_CreateSharedKey_multi(h, P(o32m), P(pk), P(sk), n),
                 lambda: L.curve25519_dh_CreateSharedKey_batch(P(o32b), P(pk), P(sk), n), o32m, o32b),
                ("sign", lambda: L.ed25519_SignMessage_multi(h, P(o64m), P(priv), P(msg), 32, n),
                 lambda: L.ed25519_SignMessage_batch(P(o64b), P(priv), P(msg), 32, n), o64m, o64b),
                ("verify", lambda: L.ed25519_VerifySignature_multi(h, P(okm), P(sig), P(pub), P(msg), 32, n),
                 lambda: L.ed25519_VerifySignature_batch(P(okb), P(sig), P(pub), P(msg), 32, n), okm, okb)):
            assert fm() == 0 and fb() == 0 and np.array_equal(a, b), name
            tm, tb = pair(fm, fb)
            print(f"{name:7s} {'page-locked' if locked else 'pageable   '} arrays, {'gather to the root' if gather else 'no gather        '}: "
                  f"*_multi({D} virtual) {tm * 1e3:7.2f} ms = {n / tm / 1e6:6.1f} M ops/s | *_batch {tb * 1e3:7.2f} ms | ratio {tb / tm:.2f} | "
                  f"copy threads {L.c25519_amd_multi_helper_threads(h)}")
        L.c25519_amd_multi_destroy(h)
    if locked:
        for a in arrays:
            assert L.c25519_amd_host_unregister(P(a)) == 0
