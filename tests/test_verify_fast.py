"""CPU tests of the exact lattice-accelerated verification path (curve25519_amd/csrc/verify_fast.cuh), run on the device
source through the host emulation (tests/host_emul/): the short-vector search against its defining properties, and the
path's verdicts against the oracle on every class of input where a shortcut could go wrong -- valid and corrupted
signatures, S >= L, garbage keys (off-curve ones must ask for the reference-order path), R encodings an encoder never
produces, keys and R's with torsion components (where a cofactored check would differ), small-order keys."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "host_emul"))
from curve25519_amd import synth  # noqa: E402
import vectors  # noqa: E402


@pytest.fixture(scope="module")
def emul():
    import build as emul_build
    lib = C.CDLL(emul_build.build())
    lib.emul_mad_overflow_count.restype = C.c_ulonglong
    yield lib
    assert lib.emul_mad_overflow_count() == 0


def run_fast(lib, sig, pk, msg):
    n = sig.shape[0]
    sig, pk = np.ascontiguousarray(sig), np.ascontiguousarray(pk)
    msg = np.ascontiguousarray(msg).reshape(n, -1)
    v, s = np.empty(n, np.int32), np.empty(n, np.int32)
    lib.emul_ed25519_verify_fast(C.c_void_p(v.ctypes.data), C.c_void_p(s.ctypes.data), C.c_void_p(sig.ctypes.data),
                                 C.c_void_p(pk.ctypes.data), C.c_void_p(msg.ctypes.data), C.c_size_t(msg.shape[1]), C.c_size_t(n))
    return v, s


def test_short_vector_properties(emul):
    hs = vectors.lattice_inputs()
    n = len(hs)
    h = np.stack([vectors.le(x, 32) for x in hs])
    rho, tau = np.empty((n, 20), np.uint8), np.empty((n, 20), np.uint8)
    neg, fits = np.empty(n, np.int32), np.empty(n, np.int32)
    trips = (C.c_ulonglong * 3)()
    emul.emul_lattice_counters(trips)
    emul.emul_lattice(C.c_void_p(rho.ctypes.data), C.c_void_p(tau.ctypes.data), C.c_void_p(neg.ctypes.data),
                      C.c_void_p(fits.ctypes.data), C.c_void_p(h.ctypes.data), C.c_size_t(n))
    emul.emul_lattice_counters(trips)
    # where the time goes: ~5 Lehmer steps (22 inner iterations each) do the work, the ten-times-dearer exact steps only
    # finish (they were 17 per element before the Lehmer phase was run down to 128 bits)
    assert trips[0] / n < 5.5 and trips[1] / n < 120 and trips[2] / n < 4, list(trips)   # (structured inputs included: 2.6; random: 0.8)
    for i, x in enumerate(hs):
        if not fits[i]:
            continue
        r, t = int.from_bytes(rho[i].tobytes(), "little"), int.from_bytes(tau[i].tobytes(), "little")
        t = -t if neg[i] else t
        assert r % 2 == 1 and 0 < r < 2**158 and abs(t) < 2**158, (x, r, t)
        assert (r * x - t) % vectors.N8L == 0, (x, r, t)                 # tau = rho * h modulo 8L, not just L
    assert fits[12:6012].all(), "a random h practically always has a short vector that fits the walk"
    # ... and the vector that comes back is as short as an odd-rho vector of this lattice gets: within a bit of the best
    # convergent (r_i, T_i) of h / 8L with an odd T_i in the longer-component norm.  (The walk starts at the wave's longest
    # vector's top digit: round 2's stopping rule returned vectors of up to 139 bits where 131 were available.)
    worst = 0
    for i in range(12, 6012):
        a, b, ta, tb, best = vectors.N8L, hs[i], 0, 1, 999
        while b:
            if tb & 1:
                best = min(best, max(b.bit_length(), abs(tb).bit_length()))
            q = a // b
            a, b, ta, tb = b, a - q * b, tb, ta - q * tb
        mine = max(int.from_bytes(rho[i].tobytes(), "little").bit_length(), int.from_bytes(tau[i].tobytes(), "little").bit_length())
        assert mine <= best + 1, (hs[i], mine, best)
        worst = max(worst, mine)
    assert worst <= 136, worst
    # h = L-1, L-2, (L-1)/2 are close to -1, -2, -1/2 modulo L but not modulo 8L: their only short vectors have an even
    # rho, so they (correctly) do not fit and go to the reference-order path; the other hand-picked values do
    assert [int(f) for f in fits[:12]] == [1, 1, 1, 1, 0, 0, 1, 1, 1, 1, 0, 1]


def test_packed_table_rows_keep_the_value(emul):
    """Table rows are stored as 256-bit integers (fe_pack_words) and unpacked by fe_from_words.  Packing is a positional sum,
    so it must be exact for every limb pattern the kernels can hand it -- fe_carry32 outputs (limb 1 may be 2^25, limb 0 up to
    2^26 + a little), product outputs (every limb up to 2^w + 2^17), all-ones and all-zero limbs -- as long as the value stays
    below 2^256; and the unpacked limbs must be 'reduced' again (fit the second-operand bound of a product) and congruent."""
    P = 2**255 - 19
    W = [26, 25] * 5
    POS = [0, 26, 51, 77, 102, 128, 153, 179, 204, 230]
    rng = np.random.default_rng(0x9ac4)
    pats = []
    pats.append([0] * 10)
    pats.append([(1 << w) - 1 for w in W])                                  # all limbs all-ones: 2^255 - 1
    pats.append([(1 << w) + (1 << 17) - 1 for w in W])                      # a product output at its bound
    pats.append([(1 << 26) + 18, 1 << 25] + [(1 << w) - 1 for w in W[2:]])  # fe_carry32's worst wrap
    pats.append([(1 << 26) - 1, 1 << 25] + [0] * 7 + [(1 << 25) - 1])
    for _ in range(2000):
        kind = rng.integers(0, 3)
        if kind == 0:
            pats.append([int(rng.integers(0, (1 << w) + (1 << 17))) for w in W])
        elif kind == 1:
            pats.append([int(rng.integers((1 << w) - 4, (1 << w) + (1 << 17))) for w in W])
        else:
            pats.append([int(rng.integers(0, 1 << w)) for w in W])
    n = len(pats)
    limbs = np.array(pats, dtype=np.uint32)
    words, back = np.zeros((n, 8), np.uint32), np.zeros((n, 10), np.uint32)
    emul.emul_fe_pack_roundtrip(C.c_void_p(words.ctypes.data), C.c_void_p(back.ctypes.data), C.c_void_p(limbs.ctypes.data), C.c_size_t(n))
    for i, l in enumerate(pats):
        value = sum(x << p for x, p in zip(l, POS))
        assert value < 2**256
        packed = sum(int(w) << (32 * k) for k, w in enumerate(words[i]))
        assert packed == value, (i, l)
        got = sum(int(x) << p for x, p in zip(back[i], POS))
        assert got % P == value % P, (i, l)
        assert all(int(back[i][j]) < (1 << W[j]) + (1 << 17) for j in range(10)), (i, list(back[i]))


def test_sigma_comb_columns_recompose_to_sigma(emul):
    """sc_comb_columns (verify_fast.cuh): the 16-bit columns the walk reads, taken apart the way ge_add_pa_comb reads them
    (top tooth = sign, the others the row index, complemented for a negative column; row idx stands for
    2^(S(T-1)) + sum of +-2^(S j)), recompose to sigma mod L -- for even and odd sigma, 0, 1, L - 1 and random values."""
    L = vectors.L
    sig = [0, 1, 2, L - 1, L - 2, 2**252, 2**252 - 1, (1 << 252) + 1] + \
          [int.from_bytes(synth.random_bytes((1, 32), 0xc0b + i)[0].tobytes(), "little") % L for i in range(56)]
    n = len(sig)
    k = np.stack([vectors.le(x, 32) for x in sig])
    dims = np.zeros(3, np.int32)
    out = np.zeros((n, 32), np.uint32)
    emul.emul_comb_columns(C.c_void_p(out.ctypes.data), C.c_void_p(dims.ctypes.data), C.c_void_p(k.ctypes.data), C.c_size_t(n))
    teeth, ncols, words = (int(x) for x in dims)
    out = out.reshape(-1)[: n * words].reshape(n, words)
    assert teeth * ncols >= 254 and words == 2 * ((ncols + 3) // 4)
    for i, x in enumerate(sig):
        total = 0
        for r in range(words // 2):
            halves = [int(out[i][2 * r]) & 0xffff, int(out[i][2 * r]) >> 16, int(out[i][2 * r + 1]) & 0xffff, int(out[i][2 * r + 1]) >> 16]
            for col, c in zip((4 * r + 3, 4 * r + 2, 4 * r + 1, 4 * r), halves):
                if col >= ncols:
                    assert c == 0
                    continue
                assert c < (1 << teeth)
                positive = (c >> (teeth - 1)) & 1
                idx = (c if positive else ~c) & ((1 << (teeth - 1)) - 1)
                row = (1 << (ncols * (teeth - 1))) + sum((1 if (idx >> j) & 1 else -1) << (ncols * j) for j in range(teeth - 1))
                total += (row if positive else -row) << col
        assert total % L == x % L, (i, x)
        assert total in (x, x + L)                      # sigma itself when odd, sigma + L when even


def test_short_vector_search_in_lock_step_waves(emul):
    """The same search with the elements grouped into waves of 8 and 64 lanes that run in lock-step, __any taken over the
    wave as on the device (tests/host_emul/valu_model.h): lanes that are done idle while others iterate, loops end when
    the LAST lane is done, word-shift hints come from other lanes.  Hand-picked h (0, 1, L-1, ... -- the ones with freak
    quotients) sit beside random ones in every wave; every element's result must equal its one-lane result."""
    hs = vectors.lattice_inputs()[:12] + [int.from_bytes(synth.random_bytes((1, 32), 0x5eed + i)[0].tobytes(), "little") % vectors.L
                                          for i in range(116)]
    n = len(hs)                                                      # 128: two waves of 64, sixteen of 8
    order = np.arange(n).reshape(2, 64).T.reshape(-1)               # interleave: every wave of 8 holds structured and random h
    h = np.stack([vectors.le(hs[i], 32) for i in order])

    def run(lanes):
        rho, tau = np.zeros((n, 20), np.uint8), np.zeros((n, 20), np.uint8)
        neg, fits = np.zeros(n, np.int32), np.zeros(n, np.int32)
        if lanes == 1:
            emul.emul_lattice(C.c_void_p(rho.ctypes.data), C.c_void_p(tau.ctypes.data), C.c_void_p(neg.ctypes.data),
                              C.c_void_p(fits.ctypes.data), C.c_void_p(h.ctypes.data), C.c_size_t(n))
        else:
            emul.emul_lattice_waves(C.c_void_p(rho.ctypes.data), C.c_void_p(tau.ctypes.data), C.c_void_p(neg.ctypes.data),
                                    C.c_void_p(fits.ctypes.data), C.c_void_p(h.ctypes.data), C.c_size_t(n), C.c_int(lanes))
        return rho, tau, neg, fits

    one = run(1)
    for lanes in (8, 64):
        got = run(lanes)
        for a, b in zip(one, got):
            assert np.array_equal(a, b), lanes
    assert one[3].sum() > 100


def test_fast_path_equals_the_oracle(emul, oracle):
    n = 1500
    sk, msg = synth.random_bytes((n, 32), 0x111), synth.random_bytes((n, 40), 0x222)
    pub, priv = oracle.ed25519_keypair(sk, threads=8)
    sig = oracle.ed25519_sign(priv, msg, threads=8)
    bsig, bmsg, _ = synth.corrupt_for_verify(sig, msg)
    bsig[5::7, 3] ^= 1
    bsig[6::7, 40] ^= 2
    for i in range(0, 60, 3):                                            # S + L: accepted by the reference
        S = int.from_bytes(bsig[i, 32:].tobytes(), "little")
        if S + vectors.L < 2**256:
            bsig[i, 32:] = vectors.le(S + vectors.L, 32)
    v, s = run_fast(emul, bsig, pub, bmsg)
    exp = oracle.ed25519_verify(bsig, pub, bmsg, threads=8)
    assert not s.any() and np.array_equal(v, exp) and 0 < exp.sum() < n
    # garbage: on-curve "keys" are decided by the fast path (and must agree), off-curve ones must ask for the slow path
    gs, gp, gm = synth.random_bytes((n, 64), 31), synth.random_bytes((n, 32), 32), synth.random_bytes((n, 40), 33)
    v, s = run_fast(emul, gs, gp, gm)
    exp = oracle.ed25519_verify(gs, gp, gm, threads=8)
    assert 0.3 * n < s.sum() < 0.7 * n and np.array_equal(v[s == 0], exp[s == 0])
    on_curve = np.array([vectors.ed_decode(int.from_bytes(k.tobytes(), "little") & (2**255 - 1), 0) is not None for k in gp])
    assert np.array_equal(s == 0, on_curve)
    # special R encodings against a valid key
    sp = vectors.special_r_encodings()
    m = sp.shape[0]
    ssig = np.concatenate([sp, bsig[:m, 32:]], axis=1)
    v, s = run_fast(emul, ssig, pub[:m], bmsg[:m])
    assert not s.any() and np.array_equal(v, oracle.ed25519_verify(ssig, pub[:m], bmsg[:m]))


def test_slow_list_path_equals_the_oracle(emul, oracle):
    """The kernel behind the walk decides the elements on the slow list with ed_verify_reference_order: the reference's
    Verify_Init + Verify_Check on the streamed table build / 4-fold walk.  Its verdicts against the oracle on valid and
    corrupted signatures, S + L, and above all garbage keys and garbage signatures -- off-curve "points", where the result
    depends on the exact order of operations."""
    n = 600
    sk, msg = synth.random_bytes((n, 32), 0x311), synth.random_bytes((n, 33), 0x322)
    pub, priv = oracle.ed25519_keypair(sk, threads=8)
    sig = oracle.ed25519_sign(priv, msg, threads=8)
    bsig, bmsg, _ = synth.corrupt_for_verify(sig, msg)
    for i in range(0, 40, 3):
        S = int.from_bytes(bsig[i, 32:].tobytes(), "little")
        if S + vectors.L < 2**256:
            bsig[i, 32:] = vectors.le(S + vectors.L, 32)
    gs, gp, gm = synth.random_bytes((n, 64), 51), synth.random_bytes((n, 32), 52), synth.random_bytes((n, 33), 53)
    for s_, p_, m_ in ((bsig, pub, bmsg), (gs, gp, gm), (bsig, gp, bmsg)):
        s_, p_, m_ = np.ascontiguousarray(s_), np.ascontiguousarray(p_), np.ascontiguousarray(m_)
        v, pt = np.empty(n, np.int32), np.empty((n, 32), np.uint8)
        emul.emul_ed25519_verify_slow(C.c_void_p(v.ctypes.data), C.c_void_p(pt.ctypes.data), C.c_void_p(s_.ctypes.data),
                                      C.c_void_p(p_.ctypes.data), C.c_void_p(m_.ctypes.data), C.c_size_t(m_.shape[1]), C.c_size_t(n))
        assert np.array_equal(v, oracle.ed25519_verify(s_, p_, m_, threads=8))
        # ... and enc(T) itself: for garbage keys every verdict is 0 whatever the path computes, the point is not
        assert np.array_equal(pt, oracle.ed25519_verify_point(s_, p_, m_))


def test_fast_path_on_torsion(emul, oracle):
    sig, pk, msg = vectors.torsion_signature_cases()
    v, s = run_fast(emul, sig, pk, msg)
    exp = oracle.ed25519_verify(sig, pk, msg)
    assert not s.any() and np.array_equal(v, exp)
    assert 0 < exp.sum() < len(exp) // 4            # roughly one of the eight torsion offsets fits per key
    lo = vectors.small_order_keys()
    gs, gm = synth.random_bytes((8, 64), 41), synth.random_bytes((8, 32), 42)
    v, s = run_fast(emul, gs, lo, gm)
    assert not s.any() and np.array_equal(v, oracle.ed25519_verify(gs, lo, gm))


def test_degenerate_but_valid_signatures(emul):
    """The committed degenerate vectors (tests/golden/degenerate_verify.npz; expected verdicts = the real reference's):
    valid signatures whose key and R are small-order points in every encoding the reference decodes, S in {0, L, 2L, 15L},
    mixed-order keys with small-order R.  The device source of the lattice path decides what it can; what it hands to the
    slow list goes through the device source of the reference-order path; the combined verdicts are the reference's."""
    d = np.load(os.path.join(ROOT, "tests", "golden", "degenerate_verify.npz"))
    sig, pk, msg, exp = (np.ascontiguousarray(d[k]) for k in ("sig", "pk", "msg", "verdict"))
    v, s = run_fast(emul, sig, pk, msg)
    n = sig.shape[0]
    vs, pt = np.empty(n, np.int32), np.empty((n, 32), np.uint8)
    emul.emul_ed25519_verify_slow(C.c_void_p(vs.ctypes.data), C.c_void_p(pt.ctypes.data), C.c_void_p(sig.ctypes.data),
                                  C.c_void_p(pk.ctypes.data), C.c_void_p(msg.ctypes.data), C.c_size_t(msg.shape[1]), C.c_size_t(n))
    assert np.array_equal(vs, exp)                                   # the reference-order path alone, on everything
    assert np.array_equal(np.where(s == 0, v, vs), exp)              # what a verification pass returns
    assert (s == 0).sum() > n // 2 and exp[s == 0].sum() > 100       # the fast path decided most of them, accepts included
