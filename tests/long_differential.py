#!/usr/bin/env python3
"""tests/long_differential.py -- opt-in long run (not collected by pytest): the HIP path against the REAL reference
(oracle/_ref, the reference's own C files compiled by oracle/Makefile) on fresh random inputs at volume, with the edge
encodings of tests/vectors.py sprinkled through every batch.  Every round: 2^20 X25519 shared keys, 2^18 key pairs +
signatures (unblinded and with a fresh blinding context), 2^18 verifications with corrupted entries, garbage keys and S + L
rewrites, three one-key two-phase batches (honest / garbage / edge key); and, for the one-operation-per-wave
kernels small calls run (csrc/coop25519.cuh), --small-calls calls of 1..2048 elements of every operation built the same way,
--mid-calls calls of 2049 .. 2^15 elements (csrc/quad25519.cuh: four lanes per element; the one-key check with a remembered comb).

    python tests/long_differential.py [--rounds 8] [--seed 1]
"""
import argparse
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
import numpy as np  # noqa: E402

import vectors  # noqa: E402
import ctypes as C  # noqa: E402

from curve25519_amd import _lib, api, synth  # noqa: E402
from oracle_lib import Reference  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=8)
ap.add_argument("--seed", type=int, default=1)
ap.add_argument("--threads", type=int, default=16)
ap.add_argument("--small-calls", type=int, default=12, help="calls of a few elements per round and operation")
ap.add_argument("--mid-calls", type=int, default=3, help="calls of 2049 .. 2^15 elements per round and operation (four lanes per element)")
args = ap.parse_args()
assert Reference.available(), "oracle/_ref is not built (make -C oracle ref)"
ref = Reference()
LIB = _lib.load()
T = args.threads
L = vectors.L


def sprinkle(arr, rng, rows):
    """overwrite random rows of a 32-byte-wide array with edge encodings"""
    idx = rng.choice(arr.shape[0], size=len(rows), replace=False)
    arr[idx] = rows
    return idx


edge32 = np.stack([np.frombuffer(vectors.le(v, 32), np.uint8) for v in
                   [0, 1, 2, 9, 2**255 - 19, 2**255 - 20, 2**255 - 18, 2**255 - 1, 2**256 - 1, 2**255, 2**255 + 9,
                    L, L - 1, L + 1, 8 * L % 2**256, 2**252, 2**254, 2**254 + 8, 325606250916557431795983626356110631294008115727848805560023387167927233504,
                    39382357235489614581723060781553021112529911719440698176882885853963445705823]])
t0 = time.time()
total = {"x25519": 0, "keypair": 0, "sign": 0, "verify": 0}
for r in range(args.rounds):
    rng = np.random.default_rng(args.seed * 1000 + r)
    n = 1 << 20
    sk = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    pk = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    sprinkle(pk, rng, np.repeat(edge32, 8, axis=0))
    sprinkle(sk, rng, np.repeat(edge32, 4, axis=0))
    got, got_sk = api.curve25519_dh_CreateSharedKey(pk, sk)
    want, want_sk = ref.x25519_shared_threaded(pk, sk, T)
    assert np.array_equal(got, want) and np.array_equal(got_sk, want_sk), f"round {r}: X25519 differs at rows {np.nonzero((got != want).any(axis=1))[0][:5]}"
    total["x25519"] += n
    m = 1 << 18
    esk = rng.integers(0, 256, (m, 32), dtype=np.uint8)
    sprinkle(esk, rng, edge32)
    mlen = int(rng.integers(0, 200))
    msg = rng.integers(0, 256, (m, mlen), dtype=np.uint8)
    pub, priv = api.ed25519_CreateKeyPair(esk)
    rpub, rpriv = ref.ed25519_keypair(esk[:4096])
    assert np.array_equal(pub[:4096], rpub) and np.array_equal(priv[:4096], rpriv), f"round {r}: keypair differs"
    total["keypair"] += 4096
    sig = api.ed25519_SignMessage(priv, msg)
    rsig = ref.ed25519_sign_threaded(priv, msg, T)
    assert np.array_equal(sig, rsig), f"round {r}: signatures differ at rows {np.nonzero((sig != rsig).any(axis=1))[0][:5]}"
    total["sign"] += m
    bsig, bmsg, bad = synth.corrupt_for_verify(sig, msg) if mlen else (sig.copy(), msg, np.zeros(m, bool))
    vpk = pub.copy()
    for i in rng.choice(m, 300, replace=False):                       # S + L where it fits: the reference accepts it
        S = int.from_bytes(bsig[i, 32:].tobytes(), "little")
        if S + L < 2**256:
            bsig[i, 32:] = np.frombuffer(vectors.le(S + L, 32), np.uint8)
    g = rng.choice(m, 2000, replace=False)
    vpk[g] = rng.integers(0, 256, (2000, 32), dtype=np.uint8)           # garbage keys: about half are off the curve
    sprinkle(vpk, rng, edge32)
    e = rng.choice(m, 400, replace=False)
    bsig[e, :32] = np.tile(edge32, (20, 1))                             # edge encodings as R
    ok = api.ed25519_VerifySignature(bsig, vpk, bmsg)
    rok = ref.ed25519_verify_threaded(bsig, vpk, bmsg, T)
    assert np.array_equal(ok, rok), f"round {r}: verdicts differ at rows {np.nonzero(ok != rok)[0][:5]}"
    total["verify"] += m
    # blinded calls: a fresh context per round (ed25519_Blinding_Init on the device), the bytes of the unblinded calls
    seed = rng.integers(0, 256, int(rng.integers(1, 130)), dtype=np.uint8)
    bctx = np.zeros(192, np.uint8)
    LIB.ed25519_Blinding_Init.restype = C.c_void_p
    assert LIB.ed25519_Blinding_Init(bctx.ctypes.data, seed.ctypes.data, len(seed)) == bctx.ctypes.data

    def blinded(lo, k):
        bp, bq, bs = np.empty((k, 32), np.uint8), np.empty((k, 64), np.uint8), np.empty((k, 64), np.uint8)
        e, q, mm = np.ascontiguousarray(esk[lo:lo + k]), np.ascontiguousarray(priv[lo:lo + k]), np.ascontiguousarray(msg[lo:lo + k])
        _lib.check(LIB.ed25519_CreateKeyPair_blinded_batch(bp.ctypes.data, bq.ctypes.data, bctx.ctypes.data, e.ctypes.data, k), "keypair blinded")
        _lib.check(LIB.ed25519_SignMessage_blinded_batch(bs.ctypes.data, q.ctypes.data, bctx.ctypes.data, mm.ctypes.data if mlen else None, mlen, k), "sign blinded")
        return bp, bq, bs

    bp, bq, bs = blinded(0, m)
    assert np.array_equal(bp, pub) and np.array_equal(bq, priv) and np.array_equal(bs, rsig), f"round {r}: blinded calls differ"
    total["blinded keypair + sign"] = total.get("blinded keypair + sign", 0) + m
    # two-phase verification, ONE key for the batch (2^16 pairs and more: two wide combs; fewer: the reference's order): an
    # honest key, a garbage one (about half are off the curve), an edge encoding (small order / non-canonical)
    for kind in range(3):
        j = int(rng.integers(0, m))
        k1 = (1 << 16) + int(rng.integers(0, 1 << 14)) if kind == 0 or r % 2 == 0 else int(rng.integers(2049, 1 << 15))
        lo = int(rng.integers(0, m - k1))
        one_priv = np.ascontiguousarray(np.broadcast_to(priv[j], (k1, 64)))
        s1 = api.ed25519_SignMessage(one_priv, msg[lo:lo + k1])
        s1b, m1, _ = synth.corrupt_for_verify(s1, msg[lo:lo + k1]) if mlen else (s1.copy(), msg[lo:lo + k1], None)
        key = pub[j].copy() if kind == 0 else rng.integers(0, 256, 32, dtype=np.uint8) if kind == 1 else edge32[int(rng.integers(0, len(edge32)))].copy()
        ctx = api.ed25519_Verify_Init(key[None, :])[0]
        v1 = api.ed25519_Verify_Check(ctx, s1b, m1)
        v2 = api.ed25519_Verify_Check(ctx, s1b, m1)                     # the remembered context: no second preparation
        rv = ref.ed25519_verify_threaded(s1b, np.ascontiguousarray(np.broadcast_to(key, (k1, 32))), m1, T)
        assert np.array_equal(v1, rv) and np.array_equal(v2, rv), f"round {r}: one-key verdicts differ (kind {kind}, n = {k1}) at rows {np.nonzero(v1 != rv)[0][:5]}"
        total["one-key verify"] = total.get("one-key verify", 0) + k1
    # calls of a few elements: other kernels (one operation per wave), same inputs' distribution
    for c in range(args.small_calls):
        k = int(rng.integers(1, 2049))
        lo = int(rng.integers(0, m - k))
        a, b = api.curve25519_dh_CreateSharedKey(pk[lo:lo + k], sk[lo:lo + k])
        assert np.array_equal(a, want[lo:lo + k]) and np.array_equal(b, want_sk[lo:lo + k]), f"round {r}: small X25519 call {c} (n = {k}) differs"
        fast, _ = api.curve25519_dh_CalculatePublicKey(sk[lo:lo + k], fast=True)
        slow, _ = api.curve25519_dh_CalculatePublicKey(sk[lo:lo + min(k, 64)])
        assert np.array_equal(fast[:len(slow)], slow) and np.array_equal(fast, ref.x25519_public(sk[lo:lo + k], fast=True)[0]), f"round {r}: small public-key call {c}"
        p2, q2 = api.ed25519_CreateKeyPair(esk[lo:lo + k])
        assert np.array_equal(p2, pub[lo:lo + k]) and np.array_equal(q2, priv[lo:lo + k]), f"round {r}: small keypair call {c} (n = {k}) differs"
        assert np.array_equal(api.ed25519_SignMessage(priv[lo:lo + k], msg[lo:lo + k]), rsig[lo:lo + k]), f"round {r}: small sign call {c} (n = {k}) differs"
        assert np.array_equal(api.ed25519_VerifySignature(bsig[lo:lo + k], vpk[lo:lo + k], bmsg[lo:lo + k]), rok[lo:lo + k]), f"round {r}: small verify call {c} (n = {k}) differs"
        bp, bq, bs = blinded(lo, k)
        assert np.array_equal(bp, pub[lo:lo + k]) and np.array_equal(bq, priv[lo:lo + k]) and np.array_equal(bs, rsig[lo:lo + k]), f"round {r}: small blinded call {c} (n = {k}) differs"
        kk = min(k, 1024)
        ctxs = api.ed25519_Verify_Init(vpk[lo:lo + kk])
        j = int(rng.integers(0, kk))
        one = api.ed25519_Verify_Check(ctxs[j], bsig[lo:lo + kk], bmsg[lo:lo + kk])
        want1 = ref.ed25519_verify_threaded(bsig[lo:lo + kk], np.ascontiguousarray(np.broadcast_to(vpk[lo + j], (kk, 32))), bmsg[lo:lo + kk], T)
        assert np.array_equal(one, want1), f"round {r}: small two-phase call {c} (n = {kk}) differs"
        for key in ("x25519", "keypair", "sign", "verify"):
            total["small " + key] = total.get("small " + key, 0) + k
    # calls of a few thousand elements: four lanes per element (csrc/quad25519.cuh), and the one-key check with a comb the thread
    # remembers from a call of 2^16 pairs with the same context
    for c in range(args.mid_calls):
        k = int(rng.integers(2049, (1 << 15) + 1))
        lo = int(rng.integers(0, m - k))
        a, b = api.curve25519_dh_CreateSharedKey(pk[lo:lo + k], sk[lo:lo + k])
        assert np.array_equal(a, want[lo:lo + k]) and np.array_equal(b, want_sk[lo:lo + k]), f"round {r}: mid-size X25519 call {c} (n = {k}) differs"
        fast, _ = api.curve25519_dh_CalculatePublicKey(sk[lo:lo + k], fast=True)
        assert np.array_equal(fast, ref.x25519_public(sk[lo:lo + k], fast=True)[0]), f"round {r}: mid-size public-key call {c} (n = {k}) differs"
        p2, q2 = api.ed25519_CreateKeyPair(esk[lo:lo + k])
        assert np.array_equal(p2, pub[lo:lo + k]) and np.array_equal(q2, priv[lo:lo + k]), f"round {r}: mid-size keypair call {c} (n = {k}) differs"
        assert np.array_equal(api.ed25519_SignMessage(priv[lo:lo + k], msg[lo:lo + k]), rsig[lo:lo + k]), f"round {r}: mid-size sign call {c} (n = {k}) differs"
        assert np.array_equal(api.ed25519_VerifySignature(bsig[lo:lo + k], vpk[lo:lo + k], bmsg[lo:lo + k]), rok[lo:lo + k]), f"round {r}: mid-size verify call {c} (n = {k}) differs"
        j = int(rng.integers(0, m))
        big = 1 << 16
        lo1 = int(rng.integers(0, m - big))
        one_priv = np.ascontiguousarray(np.broadcast_to(priv[j], (big, 64)))
        s1 = api.ed25519_SignMessage(one_priv, msg[lo1:lo1 + big])
        s1b, m1, _ = synth.corrupt_for_verify(s1, msg[lo1:lo1 + big]) if mlen else (s1.copy(), msg[lo1:lo1 + big], None)
        ctx = api.ed25519_Verify_Init(pub[j][None, :])[0]
        rv = ref.ed25519_verify_threaded(s1b, np.ascontiguousarray(np.broadcast_to(pub[j], (big, 32))), m1, T)
        assert np.array_equal(api.ed25519_Verify_Check(ctx, s1b, m1), rv), f"round {r}: one-key verdicts differ (n = 2^16)"
        off = int(rng.integers(0, big - k))
        assert np.array_equal(api.ed25519_Verify_Check(ctx, s1b[off:off + k], m1[off:off + k]), rv[off:off + k]), f"round {r}: one-key verdicts with the remembered comb differ (n = {k})"
        for key in ("x25519", "keypair", "sign", "verify", "one-key verify"):
            total["mid-size " + key] = total.get("mid-size " + key, 0) + k
    print(f"round {r}: ok (msg {mlen} B, accepted {int(ok.sum())} of {m})   {time.time() - t0:.0f} s", flush=True)
print("long differential ok:", total)
