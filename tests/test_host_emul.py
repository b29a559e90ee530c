"""CPU tests of the DEVICE SOURCE: curve25519_amd/csrc/*.cuh compiled by g++ against a C model of the gfx950
primitives (tests/host_emul/) and driven one lane at a time -- against Python big integers, the committed
fixtures (the real reference's outputs) and the oracle.  What this leaves to the GPU suite is the kernels' indexing,
LDS staging and scratch plumbing, and the asm primitives themselves (tests/test_gpu_parity.py)."""
import ctypes as C
import hashlib
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "host_emul"))
from curve25519_amd import synth  # noqa: E402
import vectors  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
KAT = json.load(open(os.path.join(GOLD, "kat.json")))
R1024 = np.load(os.path.join(GOLD, "random_1024.npz"))
sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()  # noqa: E731
vp, sz = C.c_void_p, C.c_size_t


def h2a(s):
    return np.frombuffer(bytes.fromhex(s), np.uint8).reshape(1, -1).copy()


@pytest.fixture(scope="module")
def emul():
    import build as emul_build
    lib = C.CDLL(emul_build.build())
    lib.emul_mad_overflow_count.restype = C.c_ulonglong
    for name, args in {"emul_fe_op": [vp, vp, vp, sz, C.c_int], "emul_sc_op": [vp, vp, vp, sz, C.c_int],
                       "emul_fold": [vp, vp, sz], "emul_base_table": [vp], "emul_x25519": [vp, vp, vp, sz],
                       "emul_x25519_public_fast": [vp, vp, sz], "emul_blinding_init": [vp, vp, sz],
                       "emul_ed25519_keypair": [vp, vp, vp, vp, sz], "emul_ed25519_sign": [vp, vp, vp, vp, sz, sz],
                       "emul_ed25519_verify": [vp, vp, vp, vp, vp, sz, sz], "emul_ed25519_verify_init": [vp, vp, sz]}.items():
        getattr(lib, name).argtypes = args
        getattr(lib, name).restype = None
    yield lib
    assert lib.emul_mad_overflow_count() == 0, "a v_mad_u64_u32 column wrapped 2^64: the bound contract is broken"


def ptr(a):
    return a.ctypes.data if a is not None else None


class Emul:
    def __init__(self, lib):
        self.lib = lib

    def x25519(self, pk, sk):
        sk = np.ascontiguousarray(sk).copy()
        out = np.empty_like(sk)
        self.lib.emul_x25519(ptr(out), ptr(np.ascontiguousarray(pk)) if pk is not None else None, ptr(sk), sk.shape[0])
        return out, sk

    def public_fast(self, sk):
        sk = np.ascontiguousarray(sk).copy()
        out = np.empty_like(sk)
        self.lib.emul_x25519_public_fast(ptr(out), ptr(sk), sk.shape[0])
        return out, sk

    def keypair(self, sk, blinding=None):
        n = sk.shape[0]
        pub, priv = np.empty((n, 32), np.uint8), np.empty((n, 64), np.uint8)
        self.lib.emul_ed25519_keypair(ptr(pub), ptr(priv), ptr(blinding), ptr(np.ascontiguousarray(sk)), n)
        return pub, priv

    def sign(self, priv, msg, blinding=None):
        n = priv.shape[0]
        msg = np.ascontiguousarray(msg).reshape(n, -1) if n else np.zeros((0, 0), np.uint8)
        sig = np.empty((n, 64), np.uint8)
        self.lib.emul_ed25519_sign(ptr(sig), ptr(np.ascontiguousarray(priv)), ptr(blinding), ptr(msg), msg.shape[1], n)
        return sig

    def verify(self, sig, pk, msg, point=False):
        n = sig.shape[0]
        msg = np.ascontiguousarray(msg).reshape(n, -1)
        ok, pt = np.empty(n, np.int32), np.empty((n, 32), np.uint8)
        self.lib.emul_ed25519_verify(ptr(ok), ptr(pt), ptr(np.ascontiguousarray(sig)), ptr(np.ascontiguousarray(pk)),
                                     ptr(msg), msg.shape[1], n)
        return (ok, pt) if point else ok


@pytest.fixture(scope="module")
def dev(emul):
    return Emul(emul)


def test_field_layer_against_big_integers(emul):
    pairs, a, b = vectors.field_cases()
    for op in vectors.FIELD_OPS:
        bb = b
        if op in vectors.FIELD_OPS_B_REDUCED:                # contract: the subtrahend is a reduced element
            bb = np.stack([vectors.le(y % vectors.P, 32) for _, y in pairs])
        out = np.empty((len(pairs), 32), np.uint8)
        emul.emul_fe_op(ptr(out), ptr(a), ptr(bb), len(pairs), op)
        vectors.check_field(op, out, pairs)


def test_inversions_agree_with_big_integers(emul):
    """1 / x three ways on ~3 700 patterns (vectors.inversion_cases): fe_invert (what every kernel runs), the reference's
    exponentiation x^(p-2) and the constant-time division steps of safegcd25519.cuh; 0 (and p, 2p) give 0.  The model of
    v_mad_i64_i32 reports a signed wrap (the emul fixture's overflow count)."""
    pairs, a, b = vectors.inversion_cases()
    for op in (4, 12, 13, 14):
        out = np.empty((len(pairs), 32), np.uint8)
        emul.emul_fe_op(ptr(out), ptr(a), ptr(b), len(pairs), op)
        vectors.check_field(op, out, pairs)


def test_scalar_layer_borrow_paths(emul):
    a512, b256, a, b = vectors.scalar_cases()
    for op in vectors.SCALAR_OPS:
        out = np.empty((len(a512), 32), np.uint8)
        emul.emul_sc_op(ptr(out), ptr(a), ptr(b), len(a512), op)
        vectors.check_scalar(op, out, a512, b256)


def test_fold_recodings_match_the_reference(emul):
    recs = KAT["folds"]
    k = np.concatenate([h2a(r["k"]) for r in recs])
    out = np.empty((len(recs), 128), np.uint8)
    emul.emul_fold(ptr(out), ptr(k), len(recs))
    for i, r in enumerate(recs):
        assert out[i, :32].tobytes().hex() == r["fold8"]
        assert out[i, 32:64].tobytes().hex() == r["fold8"]
        assert out[i, 64:].tobytes().hex() == r["fold4"]


def test_base_table(emul):
    tbl = np.empty((256, 3, 32), np.uint8)
    emul.emul_base_table(ptr(tbl))
    assert sha(tbl) == KAT["base_folding8_sha256"]


def test_x25519_kats(dev):
    recs = KAT["x25519"]
    shared, clamped = dev.x25519(np.concatenate([h2a(r["pk"]) for r in recs]), np.concatenate([h2a(r["sk"]) for r in recs]))
    for i, r in enumerate(recs):
        assert shared[i].tobytes().hex() == r["shared"] and clamped[i].tobytes().hex() == r["sk_clamped"], r["name"]
    recs = KAT["x25519_public"]
    sk = np.concatenate([h2a(r["sk"]) for r in recs])
    for pk, clamped in (dev.x25519(None, sk), dev.public_fast(sk)):
        for i, r in enumerate(recs):
            assert pk[i].tobytes().hex() == r["pk"] and clamped[i].tobytes().hex() == r["sk_clamped"], r["name"]


def test_ed25519_kats(dev):
    for r in KAT["ed25519"]:
        msg = np.frombuffer(bytes.fromhex(r["msg"]), np.uint8).reshape(1, -1)
        pub, priv = dev.keypair(h2a(r["sk"]))
        assert pub.tobytes().hex() == r["pk"] and priv.tobytes().hex() == r["priv"], r["name"]
        sig = dev.sign(priv, msg)
        assert sig.tobytes().hex() == r["sig"], r["name"]
        assert int(dev.verify(sig, pub, msg)[0]) == 1
    for r in KAT["ed25519_verify"]:
        msg = np.frombuffer(bytes.fromhex(r["msg"]), np.uint8).reshape(1, -1)
        assert int(dev.verify(h2a(r["sig"]), h2a(r["pk"]), msg)[0]) == r["verify"], r["name"]


def test_reference_fixture_rows(dev):
    g, m = R1024, 192
    shared, clamped = dev.x25519(g["x_pk"][:m], g["x_sk"][:m])
    assert np.array_equal(shared, g["x_shared"][:m]) and np.array_equal(clamped, g["x_sk_clamped"][:m])
    pub, priv = dev.keypair(g["ed_sk"][:m])
    assert np.array_equal(pub, g["ed_pub"][:m]) and np.array_equal(priv, g["ed_priv"][:m])
    assert np.array_equal(dev.sign(priv, g["ed_msg"][:m]), g["ed_sig"][:m])
    assert np.array_equal(dev.verify(g["v_sig"][:m], g["ed_pub"][:m], g["v_msg"][:m]), g["v_ok"][:m])


def test_verify_point_on_garbage_keys(dev, oracle):
    n = 200
    sig, pk, msg = synth.random_bytes((n, 64), 0xA501), synth.random_bytes((n, 32), 0xA502), synth.random_bytes((n, 24), 0xA503)
    ok, pt = dev.verify(sig, pk, msg, point=True)
    assert np.array_equal(pt, oracle.ed25519_verify_point(sig, pk, msg))
    assert np.array_equal(ok, oracle.ed25519_verify(sig, pk, msg))


def test_blinding_is_real_and_output_neutral(dev, emul):
    """A context derived from a seed changes the scalar the walk sees ((k + bl) mod L != k) and the starting
    point's Z, and the outputs stay byte-identical (reference ed25519_sign.c:254-259; its own harness checks the
    same equality, test/curve25519_test.c:385-394)."""
    n = 24
    sk, msg = synth.random_bytes((n, 32), 0xB101), synth.random_bytes((n, 45), 0xB102)
    pub, priv = dev.keypair(sk)
    sig = dev.sign(priv, msg)
    seen = set()
    for seed in (b"", b"x", bytes(range(64)), bytes(200)):
        ctx = np.empty(192, np.uint8)
        s = np.frombuffer(seed, np.uint8).copy() if seed else np.zeros(1, np.uint8)
        emul.emul_blinding_init(ptr(ctx), ptr(s), len(seed))
        bl = int.from_bytes(ctx[:32].tobytes(), "little")
        assert 0 < bl <= vectors.L
        t = (vectors.L - bl) % vectors.L
        seen.add(bl)
        # BP = t*B in the PE form: (y+x, y-x, 2dxy, 2) -- check it against the base table identity via keypairs:
        # (k + bl)*B + t*B == k*B is exactly what the equalities below establish for every k
        bpub, bpriv = dev.keypair(sk, blinding=ctx)
        assert np.array_equal(bpub, pub) and np.array_equal(bpriv, priv)
        assert np.array_equal(dev.sign(priv, msg, blinding=ctx), sig)
        assert t != 0
        # a context is caller-supplied bytes: the same blinding scalar written as bl + m*L (any m that fits 256 bits) is the
        # same context mathematically, and k + bl then reaches 2^256 - L and beyond -- where the signed comb's "+ L" would
        # overflow 256 bits if the sum were not reduced first.  Outputs must not move.
        for m in (14, 15):
            big = bl + m * vectors.L
            if big >= 2**256:
                continue
            ctx2 = ctx.copy()
            ctx2[:32] = vectors.le(big, 32)
            bpub, bpriv = dev.keypair(sk, blinding=ctx2)
            assert np.array_equal(bpub, pub) and np.array_equal(bpriv, priv)
            assert np.array_equal(dev.sign(priv, msg, blinding=ctx2), sig)
    assert len(seen) == 4


def test_wide_fixed_base_comb(dev, emul):
    """The wide fixed-base comb (ge25519.cuh WB_*: 13 signed teeth 20 bits apart, four tables of 4096 packed rows read through
    L2 on the device, tunable BASE_COMB = 1): table rows as the device generates them (ge_signed_comb_row) against the table
    the model builds with one addition per row; then every fixed-base operation over it -- key pairs, signatures, X25519
    public keys, with and without a blinding context, the RFC 8032 vectors and the extreme scalars included -- gives the
    bytes of the 8 x 32 LDS comb, i.e. the reference's (edp_BasePointMultiply, ed25519_sign.c:246-268)."""
    emul.emul_wide_row_check.argtypes = [C.c_int, vp, sz]
    emul.emul_wide_row_check.restype = C.c_int
    idx = np.array([0, 1, 2, 4095, 4094, 2048, 2047, 1365, 2730, 777], np.uint32)
    for table in range(4):
        assert emul.emul_wide_row_check(table, ptr(idx), len(idx)) == 0, table
    g, m = R1024, 160
    emul.emul_set_base_comb(1)
    try:
        pub, priv = dev.keypair(g["ed_sk"][:m])
        assert np.array_equal(pub, g["ed_pub"][:m]) and np.array_equal(priv, g["ed_priv"][:m])
        assert np.array_equal(dev.sign(priv, g["ed_msg"][:m]), g["ed_sig"][:m])
        for r in KAT["ed25519"]:
            msg = np.frombuffer(bytes.fromhex(r["msg"]), np.uint8).reshape(1, -1)
            pub, priv = dev.keypair(h2a(r["sk"]))
            assert pub.tobytes().hex() == r["pk"] and dev.sign(priv, msg).tobytes().hex() == r["sig"], r["name"]
        recs = KAT["x25519_public"]
        sk = np.concatenate([h2a(r["sk"]) for r in recs] + [np.zeros((1, 32), np.uint8), np.full((1, 32), 0xff, np.uint8)])
        pk, clamped = dev.public_fast(sk)
        ref_pk, ref_clamped = dev.x25519(None, sk)                     # the ladder on u = 9
        assert np.array_equal(pk, ref_pk) and np.array_equal(clamped, ref_clamped)
        for i, r in enumerate(recs):
            assert pk[i].tobytes().hex() == r["pk"], r["name"]
        # blinded: (k + bl) * B + BP over the wide comb, T of the walk's result feeding the last addition
        ctx = np.empty(192, np.uint8)
        seed = np.frombuffer(b"wide comb", np.uint8).copy()
        emul.emul_set_base_comb(0)
        emul.emul_blinding_init(ptr(ctx), ptr(seed), len(seed))
        emul.emul_set_base_comb(1)
        bpub, bpriv = dev.keypair(g["ed_sk"][:48], blinding=ctx)
        assert np.array_equal(bpub, g["ed_pub"][:48]) and np.array_equal(bpriv, g["ed_priv"][:48])
        assert np.array_equal(dev.sign(g["ed_priv"][:48], g["ed_msg"][:48], blinding=ctx), g["ed_sig"][:48])
    finally:
        emul.emul_set_base_comb(0)


def test_one_key_verification_over_two_wide_combs(dev, emul, oracle):
    """The device source of the two-phase fast path (engine.hip: k_ed25519_verify_check_wide; ge25519.cuh:
    ge_double_base_mult_wide, wb_columns<false>) on the CPU model: T = s*B + h*(-A) over the base point's wide comb and one
    built for the key, against the reference-order emulation and the oracle -- an honest key with corrupted entries and
    S + L, every small-order key with garbage and with the degenerate vectors the real reference accepts (an even h must
    become h + 1 with one -A taken off again, never h + L: a small-order key is all torsion), and an off-curve key, which
    the path must decline."""
    emul.emul_ed25519_verify_check_wide.argtypes = [vp, vp, vp, vp, sz, sz]
    emul.emul_ed25519_verify_check_wide.restype = C.c_int
    emul.emul_quad_verify_check_wide.argtypes = [vp, vp, vp, vp, sz, sz]
    emul.emul_quad_verify_check_wide.restype = C.c_int

    def wide(sig, key, msg):
        ok = np.full(sig.shape[0], -1, np.int32)
        msg = np.ascontiguousarray(msg)
        applies = emul.emul_ed25519_verify_check_wide(ptr(ok), ptr(np.ascontiguousarray(sig)), ptr(np.ascontiguousarray(key)),
                                                      ptr(msg), msg.shape[1], sig.shape[0])
        # ... and on four lock-step lanes per pair (quad::verify_check_wide_element: what calls of 2^10 .. 2^14 pairs run)
        okq = np.full(sig.shape[0], -1, np.int32)
        appq = emul.emul_quad_verify_check_wide(ptr(okq), ptr(np.ascontiguousarray(sig)), ptr(np.ascontiguousarray(key)),
                                                ptr(msg), msg.shape[1], sig.shape[0])
        assert appq == applies and np.array_equal(okq, ok), "the quad kernel's verdicts differ from the one-lane kernel's"
        return applies, ok

    n = 96
    sk, msg = synth.random_bytes((1, 32), 0xC701), synth.random_bytes((n, 19), 0xC702)
    pub, priv = dev.keypair(sk)
    sig = dev.sign(np.repeat(priv, n, axis=0), msg)
    sig[::7, 3] ^= 0x20
    msg[1::7, 18] ^= 1
    for i in range(2, n, 7):
        S = int.from_bytes(sig[i, 32:].tobytes(), "little") + vectors.L
        if S < 2**256:
            sig[i, 32:] = vectors.le(S, 32)
    keys = np.repeat(pub, n, axis=0)
    applies, ok = wide(sig, pub, msg)
    assert applies == 1 and np.array_equal(ok, dev.verify(sig, keys, msg)) and np.array_equal(ok, oracle.ed25519_verify(sig, keys, msg))
    assert 0 < ok.sum() < n
    d = np.load(os.path.join(GOLD, "degenerate_verify.npz"))
    seen = 0
    for key in np.unique(d["pk"], axis=0)[::3]:
        at = np.nonzero((d["pk"] == key).all(axis=1))[0][:40]
        gs, gm = synth.random_bytes((16, 64), 0xC703), synth.random_bytes((16, d["msg"].shape[1]), 0xC704)
        s2, m2 = np.concatenate([d["sig"][at], gs]), np.concatenate([d["msg"][at], gm])
        applies, ok = wide(s2, key.reshape(1, 32), m2)
        assert applies == 1, key.tobytes().hex()
        assert np.array_equal(ok[: len(at)], d["verdict"][at]), key.tobytes().hex()
        assert np.array_equal(ok, oracle.ed25519_verify(s2, np.repeat(key.reshape(1, 32), len(s2), axis=0), m2))
        seen += int(d["verdict"][at].sum())
    assert seen > 20
    off = next(k for k in synth.random_bytes((64, 32), 0x999) if vectors.ed_decode(int.from_bytes(k.tobytes(), "little") & (2**255 - 1), 0) is None)
    assert wide(sig, off.reshape(1, 32), msg)[0] == 0
