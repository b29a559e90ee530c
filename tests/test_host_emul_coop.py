"""CPU tests of the ONE-OPERATION-PER-WAVE device source (curve25519_amd/csrc/coop25519.cuh, coop_ops.cuh: what a call of a
few elements -- the reference's own single-call prototypes -- runs on the device).  The same source is compiled by g++ and run
as 64 (192 for the three-wave verification) lock-step lanes on the host: every lane a fiber, the DPP moves, v_permlane swaps,
wave barriers and __syncthreads of the device code as rendezvous between them (tests/host_emul/coop_wave.h).  Checked against
the committed fixtures (the real reference's outputs), the reference's degenerate-vector verdicts, and -- for the two-phase
contexts -- byte for byte against the per-lane code.  What the GPU suite adds is that the hardware's cross-lane instructions
do what the model says."""
import ctypes as C
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
vp, sz = C.c_void_p, C.c_size_t


def h2a(s):
    return np.frombuffer(bytes.fromhex(s), np.uint8).reshape(1, -1).copy()


def ptr(a):
    return a.ctypes.data if a is not None else None


def rows(a, width):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    return a.reshape(-1, width)


class Wave:
    """the per-wave operations, one emulated workgroup per element"""

    def __init__(self, lib):
        self.lib = lib

    def x25519(self, pk, sk):
        sk = rows(sk, 32).copy()
        out = np.empty_like(sk)
        self.lib.emul_coop_x25519(ptr(out), ptr(rows(pk, 32)) if pk is not None else None, ptr(sk), sk.shape[0])
        return out, sk

    def x25519_two_waves(self, pk, sk):
        sk = rows(sk, 32).copy()
        out = np.empty_like(sk)
        self.lib.emul_coop_x25519_two_waves(ptr(out), ptr(rows(pk, 32)), ptr(sk), sk.shape[0])
        return out, sk

    def public_fast(self, sk, wide=1):
        sk = rows(sk, 32).copy()
        out = np.empty_like(sk)
        self.lib.emul_coop_public_fast(ptr(out), ptr(sk), sk.shape[0], wide)
        return out, sk

    def keypair(self, sk, blinding=None, wide=1):
        sk = rows(sk, 32)
        pub, priv = np.empty((sk.shape[0], 32), np.uint8), np.empty((sk.shape[0], 64), np.uint8)
        self.lib.emul_coop_keypair(ptr(pub), ptr(priv), ptr(blinding), ptr(sk), sk.shape[0], wide)
        return pub, priv

    def sign(self, priv, msg, blinding=None, wide=1):
        priv = rows(priv, 64)
        n = priv.shape[0]
        msg = np.ascontiguousarray(msg, dtype=np.uint8).reshape(n, -1)
        sig = np.empty((n, 64), np.uint8)
        self.lib.emul_coop_sign(ptr(sig), ptr(priv), ptr(blinding), ptr(msg) if msg.shape[1] else None, msg.shape[1], n, wide)
        return sig

    def verify(self, sig, pk, msg, cap_bits=0):
        """the three-wave lattice path: (verdicts, went-on-the-slow-list); a listed element's verdict stays -1"""
        sig, pk = rows(sig, 64), rows(pk, 32)
        n = sig.shape[0]
        msg = np.ascontiguousarray(msg, dtype=np.uint8).reshape(n, -1)
        ok, slow = np.full(n, -1, np.int32), np.zeros(n, np.int32)
        self.lib.emul_coop_verify_three_waves(ptr(ok), ptr(slow), ptr(sig), ptr(pk), ptr(msg) if msg.shape[1] else None, msg.shape[1], n, cap_bits)
        return ok, slow

    def verify_init(self, pk):
        pk = rows(pk, 32)
        ctx = np.empty((pk.shape[0], 2080), np.uint8)
        self.lib.emul_coop_verify_init(ptr(ctx), ptr(pk), pk.shape[0])
        return ctx

    def verify_check(self, ctx, sig, msg):
        sig = rows(sig, 64)
        n = sig.shape[0]
        msg = np.ascontiguousarray(msg, dtype=np.uint8).reshape(n, -1)
        ok = np.full(n, -1, np.int32)
        self.lib.emul_coop_verify_check(ptr(ok), ptr(np.ascontiguousarray(ctx)), ptr(sig), ptr(msg) if msg.shape[1] else None, msg.shape[1], n)
        return ok


@pytest.fixture(scope="module")
def lib():
    import build as emul_build
    lib = C.CDLL(emul_build.build())
    lib.emul_mad_overflow_count.restype = C.c_ulonglong
    lib.emul_coop_sync_points.restype = C.c_ulonglong
    for name, args in {"emul_coop_x25519": [vp, vp, vp, sz], "emul_coop_x25519_two_waves": [vp, vp, vp, sz],
                       "emul_coop_public_fast": [vp, vp, sz, C.c_int],
                       "emul_coop_keypair": [vp, vp, vp, vp, sz, C.c_int], "emul_coop_sign": [vp, vp, vp, vp, sz, sz, C.c_int],
                       "emul_coop_blinding_init": [vp, vp, sz], "emul_coop_verify_init": [vp, vp, sz],
                       "emul_coop_verify_check": [vp, vp, vp, vp, sz, sz],
                       "emul_coop_verify_three_waves": [vp, vp, vp, vp, vp, sz, sz, C.c_int],
                       "emul_blinding_init": [vp, vp, sz], "emul_ed25519_verify_init": [vp, vp, sz],
                       "emul_ed25519_verify": [vp, vp, vp, vp, vp, sz, sz],
                       "emul_ed25519_verify_fast": [vp, vp, vp, vp, vp, sz, sz]}.items():
        getattr(lib, name).argtypes = args
        getattr(lib, name).restype = None
    yield lib
    assert lib.emul_mad_overflow_count() == 0, "a v_mad_u64_u32 column wrapped 2^64: the bound contract is broken"


@pytest.fixture(scope="module")
def wave(lib):
    return Wave(lib)


def test_x25519_per_wave_gives_the_reference_bytes(wave, lib):
    """RFC 7748 and the edge public keys of SURVEY 3.5 (0, 1, p - 1, p, p + 1, 2^255 - 1, 2^256 - 1, ...) through the per-wave
    ladder, inversion and encoding; the base-point ladder and CalculatePublicKey_fast over both fixed-base combs; rows of the
    reference's 1024-row fixture."""
    before = lib.emul_coop_sync_points()
    recs = KAT["x25519"]
    shared, clamped = wave.x25519(np.concatenate([h2a(r["pk"]) for r in recs]), np.concatenate([h2a(r["sk"]) for r in recs]))
    for i, r in enumerate(recs):
        assert shared[i].tobytes().hex() == r["shared"] and clamped[i].tobytes().hex() == r["sk_clamped"], r["name"]
    recs = KAT["x25519_public"]
    sk = np.concatenate([h2a(r["sk"]) for r in recs])
    for pk, clamped in (wave.x25519(None, sk), wave.public_fast(sk, wide=1), wave.public_fast(sk, wide=0)):
        for i, r in enumerate(recs):
            assert pk[i].tobytes().hex() == r["pk"] and clamped[i].tobytes().hex() == r["sk_clamped"], r["name"]
    g, m = R1024, 6
    shared, clamped = wave.x25519(g["x_pk"][:m], g["x_sk"][:m])
    assert np.array_equal(shared, g["x_shared"][:m]) and np.array_equal(clamped, g["x_sk_clamped"][:m])
    assert lib.emul_coop_sync_points() > before                       # the lanes did meet (the scheduler ran, not a one-lane stub)


def test_x25519_on_two_waves_gives_the_reference_bytes(wave):
    """The ladder step in two product levels (coop25519.cuh: one wave the differential addition with x1 times the sum point carried
    along, the other the doubling; 128 lock-step lanes, a workgroup barrier per step): RFC 7748, every edge public key (0, 1, p - 1, p, p + 1, 2^255 - 1, 2^256 - 1: x1 = 0 makes
    the scaled point vanish), rows of the reference's fixture -- and byte for byte the one-wave kernel's results."""
    recs = KAT["x25519"]
    pk, sk = np.concatenate([h2a(r["pk"]) for r in recs]), np.concatenate([h2a(r["sk"]) for r in recs])
    shared, clamped = wave.x25519_two_waves(pk, sk)
    for i, r in enumerate(recs):
        assert shared[i].tobytes().hex() == r["shared"] and clamped[i].tobytes().hex() == r["sk_clamped"], r["name"]
    g, m = R1024, 6
    shared, clamped = wave.x25519_two_waves(g["x_pk"][10:10 + m], g["x_sk"][10:10 + m])
    assert np.array_equal(shared, g["x_shared"][10:10 + m]) and np.array_equal(clamped, g["x_sk_clamped"][10:10 + m])
    one, _ = wave.x25519(g["x_pk"][10:12], g["x_sk"][10:12])
    assert np.array_equal(one, shared[:2])


def test_ed25519_per_wave_gives_the_reference_bytes(wave):
    """RFC 8032 / the reference's own vectors: key pair and signature per wave over the wide comb (every vector) and the LDS
    comb's tables (a few), verification by the three-wave lattice path and by the two-phase per-wave kernels."""
    for j, r in enumerate(KAT["ed25519"]):
        msg = np.frombuffer(bytes.fromhex(r["msg"]), np.uint8).reshape(1, -1)
        pub, priv = wave.keypair(h2a(r["sk"]))
        assert pub.tobytes().hex() == r["pk"] and priv.tobytes().hex() == r["priv"], r["name"]
        sig = wave.sign(priv, msg)
        assert sig.tobytes().hex() == r["sig"], r["name"]
        if j % 6 == 0:
            p0, q0 = wave.keypair(h2a(r["sk"]), wide=0)
            assert np.array_equal(p0, pub) and np.array_equal(q0, priv) and np.array_equal(wave.sign(priv, msg, wide=0), sig), r["name"]
        if j % 2 == 0:
            ok, slow = wave.verify(sig, pub, msg)
            assert int(ok[0]) == 1 and int(slow[0]) == 0, r["name"]
        if j % 5 == 0:
            assert int(wave.verify_check(wave.verify_init(pub)[0], sig, msg)[0]) == 1, r["name"]
    for r in KAT["ed25519_verify"]:
        msg = np.frombuffer(bytes.fromhex(r["msg"]), np.uint8).reshape(1, -1)
        ok, slow = wave.verify(h2a(r["sig"]), h2a(r["pk"]), msg)
        assert int(slow[0]) == 1 or int(ok[0]) == r["verify"], r["name"]
        assert int(wave.verify_check(wave.verify_init(h2a(r["pk"]))[0], h2a(r["sig"]), msg)[0]) == r["verify"], r["name"]
    g, m = R1024, 4
    pub, priv = wave.keypair(g["ed_sk"][:m])
    assert np.array_equal(pub, g["ed_pub"][:m]) and np.array_equal(priv, g["ed_priv"][:m])
    assert np.array_equal(wave.sign(priv, g["ed_msg"][:m]), g["ed_sig"][:m])
    lo = int(np.nonzero(g["v_ok"] == 0)[0][0]) - 2                     # a window with a rejected entry in it
    ok, slow = wave.verify(g["v_sig"][lo:lo + m], g["ed_pub"][lo:lo + m], g["v_msg"][lo:lo + m])
    assert np.array_equal(ok, g["v_ok"][lo:lo + m]) and not slow.any()


def test_degenerate_vectors_through_the_three_wave_path_and_the_two_phase_kernels(wave, lib):
    """tests/golden/degenerate_verify.npz (small-order and mixed-order keys, small-order R in every encoding, S in {0, L, 2L,
    15L}, off-curve keys; verdicts = the real reference's): a sample of every label through the three-wave lattice path --
    elements with an off-curve key must go on the slow list exactly where the per-lane path sends them, every other verdict is
    the reference's -- and through Verify_Init / Verify_Check per wave, whose contexts are the per-lane code's bytes."""
    d = np.load(os.path.join(GOLD, "degenerate_verify.npz"))
    pick = np.arange(0, 1024, 27)                                       # 38 vectors across the file's labels
    sig, pk, msg, exp = (np.ascontiguousarray(d[k][pick]) for k in ("sig", "pk", "msg", "verdict"))
    n = len(pick)
    ok, slow = wave.verify(sig, pk, msg)
    lane_ok, lane_slow = np.full(n, -1, np.int32), np.zeros(n, np.int32)
    lib.emul_ed25519_verify_fast(ptr(lane_ok), ptr(lane_slow), ptr(sig), ptr(pk), ptr(msg), msg.shape[1], n)
    assert np.array_equal(slow, lane_slow)
    decided = slow == 0
    assert decided.sum() >= n // 2 and np.array_equal(ok[decided], exp[decided])
    assert (ok[~decided] == -1).all()                                   # the slow list's elements are k_ed25519_verify_slow's to decide
    # an over-long short vector takes the same exit (the test knob's cap: inside the typical 127-131 bits)
    ok2, slow2 = wave.verify(sig[decided][:6], pk[decided][:6], msg[decided][:6], cap_bits=120)
    assert slow2.sum() >= 1 and np.array_equal(ok2[slow2 == 0], exp[decided][:6][slow2 == 0])
    # two-phase: contexts byte for byte the per-lane kernel's; verdicts the reference's, off-curve keys included
    sub = np.arange(0, n, 3)
    ctx = wave.verify_init(pk[sub])
    lane_ctx = np.empty_like(ctx)
    lib.emul_ed25519_verify_init(ptr(lane_ctx), ptr(np.ascontiguousarray(pk[sub])), len(sub))
    assert np.array_equal(ctx, lane_ctx)
    for j, i in enumerate(sub):
        assert int(wave.verify_check(ctx[j], sig[i:i + 1], msg[i:i + 1])[0]) == int(exp[i]), int(pick[i])


def test_blinding_context_and_blinded_calls_per_wave(wave, lib):
    """ed25519_Blinding_Init by one wave gives the per-lane kernel's 192 bytes; key pairs and signatures per wave with a context
    (the blinded scalar and starting point through the wide comb, + BP at the end) are the unblinded bytes."""
    sk, msg = synth.random_bytes((3, 32), 0xC101), synth.random_bytes((3, 45), 0xC102)
    pub, priv = wave.keypair(sk)
    sig = wave.sign(priv, msg)
    for seed in (b"", bytes(range(64)), bytes(131)):
        s = np.frombuffer(seed, np.uint8).copy() if seed else np.zeros(1, np.uint8)
        ctx, lane_ctx = np.empty(192, np.uint8), np.empty(192, np.uint8)
        lib.emul_coop_blinding_init(ptr(ctx), ptr(s), len(seed))
        lib.emul_blinding_init(ptr(lane_ctx), ptr(s), len(seed))
        assert np.array_equal(ctx, lane_ctx)
        bpub, bpriv = wave.keypair(sk, blinding=ctx)
        assert np.array_equal(bpub, pub) and np.array_equal(bpriv, priv)
        assert np.array_equal(wave.sign(priv, msg, blinding=ctx), sig)
    big = int.from_bytes(ctx[:32].tobytes(), "little") + 14 * vectors.L  # the same context written as bl + 14 L
    if big < 2**256:
        ctx2 = ctx.copy()
        ctx2[:32] = vectors.le(big, 32)
        assert np.array_equal(wave.sign(priv, msg, blinding=ctx2), sig)
