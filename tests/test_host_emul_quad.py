"""CPU tests of the FOUR-LANES-PER-ELEMENT device source (curve25519_amd/csrc/quad25519.cuh: what batches of 2^12 .. 2^14 elements
run on the device).  The same source is compiled by g++ and run as 64 lock-step lanes on the host -- 16 elements per wave, the
v_mov_b32_dpp quad_perm exchanges between a quad's lanes as rendezvous (tests/host_emul/coop_wave.h) -- against the committed
fixtures (the real reference's outputs).  What the GPU suite adds is that the hardware's quad_perm does what the model says."""
import ctypes as C
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "host_emul"))

GOLD = os.path.join(ROOT, "tests", "golden")
KAT = json.load(open(os.path.join(GOLD, "kat.json")))
R1024 = np.load(os.path.join(GOLD, "random_1024.npz"))
vp, sz = C.c_void_p, C.c_size_t


def h2a(s):
    return np.frombuffer(bytes.fromhex(s), np.uint8).reshape(1, -1).copy()


def ptr(a):
    return a.ctypes.data if a is not None else None


@pytest.fixture(scope="module")
def lib():
    import build as emul_build
    lib = C.CDLL(emul_build.build())
    lib.emul_mad_overflow_count.restype = C.c_ulonglong
    lib.emul_coop_sync_points.restype = C.c_ulonglong
    for name, args in {"emul_quad_x25519": [vp, vp, vp, sz], "emul_quad_verify": [vp, vp, vp, vp, vp, sz, sz],
                       "emul_quad_keypair": [vp, vp, vp, sz], "emul_quad_sign": [vp, vp, vp, sz, sz], "emul_quad_public_fast": [vp, vp, sz],
                       "emul_ed25519_verify_fast": [vp, vp, vp, vp, vp, sz, sz]}.items():
        getattr(lib, name).argtypes = args
        getattr(lib, name).restype = None
    yield lib
    assert lib.emul_mad_overflow_count() == 0, "a v_mad_u64_u32 column wrapped 2^64: the bound contract is broken"


def quad_x25519(lib, pk, sk):
    sk = np.ascontiguousarray(sk, dtype=np.uint8).reshape(-1, 32).copy()
    out = np.empty_like(sk)
    pk = None if pk is None else np.ascontiguousarray(pk, dtype=np.uint8).reshape(-1, 32)
    lib.emul_quad_x25519(ptr(out), ptr(pk), ptr(sk), sk.shape[0])
    return out, sk


def test_x25519_on_quads_gives_the_reference_bytes(lib):
    """RFC 7748, the reference's test inputs and the edge public keys of SURVEY 3.5 (0, 1, p - 1, p, p + 1, 2^255 - 1, 2^256 - 1,
    ...: a zero Z must come out as zero bytes) -- more records than one wave's 16 elements, so a partly filled wave runs too;
    the base-point ladder (curve25519_dh_CalculatePublicKey: level 3 is a multiplication by 9); rows of the reference's
    1024-row fixture; the clamped keys written back."""
    before = lib.emul_coop_sync_points()
    recs = KAT["x25519"]
    shared, clamped = quad_x25519(lib, np.concatenate([h2a(r["pk"]) for r in recs]), np.concatenate([h2a(r["sk"]) for r in recs]))
    for i, r in enumerate(recs):
        assert shared[i].tobytes().hex() == r["shared"] and clamped[i].tobytes().hex() == r["sk_clamped"], r["name"]
    recs = KAT["x25519_public"]
    pk, clamped = quad_x25519(lib, None, np.concatenate([h2a(r["sk"]) for r in recs]))
    for i, r in enumerate(recs):
        assert pk[i].tobytes().hex() == r["pk"] and clamped[i].tobytes().hex() == r["sk_clamped"], r["name"]
    g, m = R1024, 21                                                   # one full wave and five elements of the next
    shared, clamped = quad_x25519(lib, g["x_pk"][100:100 + m], g["x_sk"][100:100 + m])
    assert np.array_equal(shared, g["x_shared"][100:100 + m]) and np.array_equal(clamped, g["x_sk_clamped"][100:100 + m])
    assert lib.emul_coop_sync_points() > before                       # the lanes did meet (the scheduler ran, not a one-lane stub)


def test_x25519_on_quads_in_place(lib):
    """`shared` may alias `pk` (curve25519_dh.c:104,150: the base point is copied first, the output written last)."""
    g = R1024
    buf = g["x_pk"][:5].copy()
    sk = g["x_sk"][:5].copy()
    lib.emul_quad_x25519(ptr(buf), ptr(buf), ptr(sk), 5)
    assert np.array_equal(buf, g["x_shared"][:5]) and np.array_equal(sk, g["x_sk_clamped"][:5])


def quad_verify(lib, sig, pk, msg):
    sig, pk = np.ascontiguousarray(sig, dtype=np.uint8).reshape(-1, 64), np.ascontiguousarray(pk, dtype=np.uint8).reshape(-1, 32)
    n = sig.shape[0]
    msg = np.ascontiguousarray(msg, dtype=np.uint8).reshape(n, -1)
    ok, slow = np.full(n, -1, np.int32), np.zeros(n, np.int32)
    lib.emul_quad_verify(ptr(ok), ptr(slow), ptr(sig), ptr(pk), ptr(msg) if msg.shape[1] else None, msg.shape[1], n)
    return ok, slow


def test_verification_walk_on_quads_gives_the_reference_verdicts(lib):
    """The lattice path's walk by quads (quad::walk_is_neutral: additions in two product levels with one field of the table row
    per lane, doublings as a level of squarings and a level of products): RFC 8032 and the reference's own vectors, a window of
    the 1024-row fixture with rejected entries in it (16 elements per wave walk together from the wave's top digit), and the
    one-lane walk's verdicts and slow-list decisions element for element."""
    for r in KAT["ed25519"]:
        msg = np.frombuffer(bytes.fromhex(r["msg"]), np.uint8).reshape(1, -1)
        ok, slow = quad_verify(lib, h2a(r["sig"]), h2a(r["pk"]), msg)
        assert ok[0] == 1 and slow[0] == 0, r["name"]
        if msg.shape[1]:
            bad = msg.copy(); bad[0, 0] ^= 1
            assert quad_verify(lib, h2a(r["sig"]), h2a(r["pk"]), bad)[0][0] == 0, r["name"]
    g, m = R1024, 37
    lo = int(np.nonzero(g["v_ok"] == 0)[0][0]) - 5                     # a window with rejected entries in it
    ok, slow = quad_verify(lib, g["v_sig"][lo:lo + m], g["ed_pub"][lo:lo + m], g["v_msg"][lo:lo + m])
    assert np.array_equal(ok, g["v_ok"][lo:lo + m]) and not slow.any() and (ok == 0).any()


def test_degenerate_vectors_through_the_quad_walk(lib):
    """tests/golden/degenerate_verify.npz (small-order and mixed-order keys, small-order R in every encoding, S in {0, L, 2L, 15L},
    off-curve keys; verdicts = the real reference's): a sample of every label.  The quad walk starts from the neutral element and
    adds neutral rows for zero digits -- exactly the inputs where a formula that is not complete would show."""
    d = np.load(os.path.join(GOLD, "degenerate_verify.npz"))
    pick = np.arange(0, 1024, 13)
    sig, pk, msg, exp = (np.ascontiguousarray(d[k][pick]) for k in ("sig", "pk", "msg", "verdict"))
    n = len(pick)
    ok, slow = quad_verify(lib, sig, pk, msg)
    lane_ok, lane_slow = np.full(n, -1, np.int32), np.zeros(n, np.int32)
    lib.emul_ed25519_verify_fast(ptr(lane_ok), ptr(lane_slow), ptr(sig), ptr(pk), ptr(msg), msg.shape[1], n)
    assert np.array_equal(slow, lane_slow)
    decided = slow == 0
    assert decided.sum() >= n // 2 and np.array_equal(ok[decided], exp[decided]) and np.array_equal(ok[decided], lane_ok[decided])
    assert exp[decided].sum() > 10 and (exp[decided] == 0).sum() > 0


def test_fixed_base_operations_on_quads_give_the_reference_bytes(lib):
    """ed25519_CreateKeyPair, ed25519_SignMessage and curve25519_dh_CalculatePublicKey_fast with the walk over the wide comb on quads
    (quad::base_mult_wide: 20 additions of two product levels from the neutral element, the doublings as a level of squarings and
    one of products, one field of the packed row -- the fourth the constant 2 in the row's padding -- per lane; inversion,
    encoding, the last hash and S in the same call): RFC 8032 and the reference's vectors (message lengths 0 .. 1023 bytes), 21
    rows of the 1024-row fixture (one full wave and five elements of the next), the public keys of the X25519 vectors."""
    before = lib.emul_coop_sync_points()
    for r in KAT["ed25519"]:
        sk, msg = h2a(r["sk"]), np.frombuffer(bytes.fromhex(r["msg"]), np.uint8).reshape(1, -1).copy()
        pub, priv = np.empty((1, 32), np.uint8), np.empty((1, 64), np.uint8)
        lib.emul_quad_keypair(ptr(pub), ptr(priv), ptr(sk), 1)
        assert pub.tobytes().hex() == r["pk"] and priv.tobytes().hex() == r["priv"], r["name"]
        sig = np.empty((1, 64), np.uint8)
        lib.emul_quad_sign(ptr(sig), ptr(priv), ptr(msg) if msg.shape[1] else None, msg.shape[1], 1)
        assert sig.tobytes().hex() == r["sig"], r["name"]
    g, m = R1024, 21
    sk = np.ascontiguousarray(g["ed_sk"][300:300 + m])
    pub, priv = np.empty((m, 32), np.uint8), np.empty((m, 64), np.uint8)
    lib.emul_quad_keypair(ptr(pub), ptr(priv), ptr(sk), m)
    assert np.array_equal(pub, g["ed_pub"][300:300 + m]) and np.array_equal(priv, g["ed_priv"][300:300 + m])
    msg = np.ascontiguousarray(g["ed_msg"][300:300 + m])
    sig = np.empty((m, 64), np.uint8)
    lib.emul_quad_sign(ptr(sig), ptr(priv), ptr(msg), msg.shape[1], m)
    assert np.array_equal(sig, g["ed_sig"][300:300 + m])
    recs = KAT["x25519_public"]
    sk = np.concatenate([h2a(r["sk"]) for r in recs])
    pk = np.empty_like(sk)
    lib.emul_quad_public_fast(ptr(pk), ptr(sk), len(recs))
    for i, r in enumerate(recs):
        assert pk[i].tobytes().hex() == r["pk"] and sk[i].tobytes().hex() == r["sk_clamped"], r["name"]
    assert lib.emul_coop_sync_points() > before
