// curve25519_amd/csrc/verify_fast.cuh -- exact Ed25519 verification with half the doublings, for keys that are on
// the curve.
//
// The reference (source/ed25519_verify.c:287-313) accepts iff enc(T) == R bytes with T = s*B + h*Q, Q = -A decompressed
// WITHOUT validation; it pays 255 doublings for the 253-bit h (192 building the 4-fold table, 63 walking it), and for
// a key used once no comb shape changes that count.  What does: a short vector of the lattice
// {(rho, tau): tau = rho*h mod 8L}.  E(F_p) has order 8L, so [rho*h]Q = [tau]Q for ANY curve point Q, torsion included,
// and if rho is odd (and 0 < rho < L) then gcd(rho, 8L) = 1 and multiplication by rho is a bijection of the group:
//         T == R   <=>   [rho](s*B + h*Q - R) == O   <=>   [rho*s mod L]B + [tau]Q + [rho](-R) == O
// with rho, tau of ~128 bits: ~134 doublings instead of 255.  When Q is on the curve the reference's own formulas are
// the complete group law (a = -1 is a square, d is not), so its T is the group's T, and enc() is injective on curve
// points: "R bytes decode canonically to a curve point R and the right-hand side holds" IS the reference's verdict,
// for every s (S >= L included: B has order L) and every torsion component of A or R.  Everything else -- a key that
// does not decompress onto the curve, a lattice vector longer than the walk's capacity -- is left to the
// reference-order path (k_ed25519_verify_slow), selected per element.
// Public data only: nothing here needs to be constant-time.
#pragma once
#include "ge25519.cuh"
#include "lanes.cuh"
#include "sc25519.cuh"
#include "sha512.cuh"

namespace c25519 {

// ---- lattice reduction -------------------------------------------------------------------------------------------------
// Euclid on (8L, h) carrying the cofactor of h.  Every vector (r, T) kept has r >= 0 and r = T*h (mod 8L) with T a
// signed integer; every step is a unimodular combination of the two current vectors, so they always form a basis of
// the lattice and their T's are never both even.  Two phases, both on 32-bit words:
//   * Lehmer steps: the leading 62 bits of both remainders go through a shift-subtract Euclid in two registers that
//     accumulates a 2x2 matrix with entries below 2^31 (about 28 bits of progress), which is then applied exactly to
//     the full vectors (a handful of v_mad_u64_u32 chains).  The leading-bits quotients need not be the true ones: any
//     unimodular matrix keeps the invariants, only how fast the vectors shrink depends on them;
//   * exact shift-subtract steps finish: until the smaller remainder is below 2^128, and then the other vector only
//     until its remainder is below 2^129, which keeps its cofactor as short as the lattice allows.
// Whatever comes out is checked (odd, short enough) before it is used.
constexpr int LAT_CAP_BITS = 158;          // what the 40-digit signed walk can take.  The typical vector has 127-131 bits; the tail halves per
                                            // bit (142 bits, round 2's capacity, was exceeded by 6 random h in 2^20), so a random h practically
                                            // never asks for the reference-order path; a wave walks from ITS longest vector's top digit anyway
#ifndef C25519_LAT_COUNT
#define C25519_LAT_COUNT(what)          // the host emulation counts loop trips here (tests/host_emul)
#endif
constexpr int LAT_LEHMER_STOP = 128;        // Lehmer steps bring the smaller remainder down to this many bits; the
                                            // exact steps (ten times dearer per bit) only finish: 128 / 129 bits
#ifndef C25519_LAT_BALANCED_STOP
#define C25519_LAT_BALANCED_STOP 1          // A/B switch: 0 = round 2's stopping rule (both remainders down to 128 / 129 bits)
#endif
constexpr int LAT_R = 8, LAT_T = 6;        // words of a remainder (unsigned) / of a cofactor (two's complement)

template <int W>
C25519_DEV int bitlen_words(const u32 (&a)[W])
{
    int n = 0;
#pragma unroll
    for (int i = 0; i < W; i++) n = a[i] ? 32 * i + (32 - __builtin_clz(a[i])) : n;
    return n;
}
C25519_DEV int bitlen64(u64 x) { return x ? 64 - __builtin_clzll(x) : 0; }
template <int W>
C25519_DEV bool geq_words(const u32 (&a)[W], const u32 (&b)[W])            // a >= b, unsigned
{
    u32 borrow = 0;
#pragma unroll
    for (int i = 0; i < W; i++) {
        const u64 d = (u64)a[i] - b[i] - borrow;
        borrow = (u32)(d >> 63);
    }
    return borrow == 0;
}
// r = a << k for 0 <= k < 32 * W (bits shifted out are lost).  any_word_shift: wave-uniform hint that some lane has k >= 32.
template <int W>
C25519_DEV void shl_var(u32 (&r)[W], const u32 (&a)[W], int k, bool any_word_shift)
{
    const int b = k & 31;
    u32 t[W];
#pragma unroll
    for (int i = W - 1; i >= 0; i--) {
        const u64 pair = ((u64)a[i] << 32) | (i ? a[i - 1] : 0u);
        t[i] = (u32)(pair >> (32 - b));
    }
    if (any_word_shift) {
        const int ws = k >> 5;
#pragma unroll
        for (int stage = 4; stage >= 1; stage >>= 1) {
            if (stage >= W) continue;
            const bool on = (ws & stage) != 0;
#pragma unroll
            for (int i = W - 1; i >= 0; i--) t[i] = on ? (i >= stage ? t[i - stage] : 0u) : t[i];
        }
    }
#pragma unroll
    for (int i = 0; i < W; i++) r[i] = t[i];
}
// the 62 bits of a below bit position `top` (top >= 62), or a itself when top < 62
template <int W>
C25519_DEV u64 leading62(const u32 (&a)[W], int top, bool any_word_shift)
{
    const int sh = top > 62 ? top - 62 : 0;
    u32 t[W];
    // right shift by sh: bits, then words
    const int b = sh & 31;
#pragma unroll
    for (int i = 0; i < W; i++) {
        const u64 pair = ((u64)(i + 1 < W ? a[i + 1] : 0u) << 32) | a[i];
        t[i] = (u32)(pair >> b);
    }
    if (any_word_shift) {
        const int ws = sh >> 5;
#pragma unroll
        for (int stage = 4; stage >= 1; stage >>= 1) {
            if (stage >= W) continue;
            const bool on = (ws & stage) != 0;
#pragma unroll
            for (int i = 0; i < W; i++) t[i] = on ? (i + stage < W ? t[i + stage] : 0u) : t[i];
        }
    }
    return ((u64)t[1] << 32) | t[0];
}
// out = A*x - B*y modulo 2^(32 W)   (A, B < 2^32; two's complement when the operands are sign-extended)
template <int W>
C25519_DEV void lin_comb(u32 (&out)[W], const u32 (&x)[W], u32 A, const u32 (&y)[W], u32 B)
{
    u32 p[W], q[W];
    u64 c = 0;
#pragma unroll
    for (int i = 0; i < W; i++) { c += (u64)x[i] * A; p[i] = (u32)c; c >>= 32; }
    c = 0;
#pragma unroll
    for (int i = 0; i < W; i++) { c += (u64)y[i] * B; q[i] = (u32)c; c >>= 32; }
    u32 borrow = 0;
#pragma unroll
    for (int i = 0; i < W; i++) {
        const u64 d = (u64)p[i] - q[i] - borrow;
        out[i] = (u32)d;
        borrow = (u32)(d >> 63);
    }
}
template <int W>
C25519_DEV void negate_words(u32 (&a)[W])
{
    u32 carry = 1;
#pragma unroll
    for (int i = 0; i < W; i++) {
        const u64 s = (u64)(~a[i]) + carry;
        a[i] = (u32)s;
        carry = (u32)(s >> 32);
    }
}

// 8L as eight 32-bit words
C25519_DEV void lat_modulus(u32 (&n)[8])
{
    n[0] = K_L[0] << 3;
#pragma unroll
    for (int i = 1; i < 8; i++) n[i] = (K_L[i] << 3) | (K_L[i - 1] >> 29);
}

#ifndef C25519_LAT_QUOTIENT_STEPS
#define C25519_LAT_QUOTIENT_STEPS 1        // A/B switch: 0 = the shift-subtract inner loop of round 2
#endif
// an estimate of xa / xb (xb != 0) that never exceeds the true quotient, and is at least 1 when xa >= xb: the top 32 bits of
// xa against the same bits of xb plus one, divided in single precision and scaled down by more than the rounding errors add up to
C25519_DEV u32 lat_quot_est(u64 xa, u64 xb)
{
    const int la = bitlen64(xa);
    const int t = la > 32 ? la - 32 : 0;
    const u32 ha = (u32)(xa >> t), hb = (u32)(xb >> t);
    const u32 q = (u32)(fast_div((float)ha, (float)hb + 1.0f) * 0.999999f);
    return xa >= xb ? (q ? q : 1u) : 0u;
}

// one Lehmer step on the vectors (r0, t0), (r1, t1).  Returns false for a lane that made no progress.
C25519_DEV bool lat_lehmer_step(u32 (&r0)[LAT_R], u32 (&t0)[LAT_T], u32 (&r1)[LAT_R], u32 (&t1)[LAT_T], bool active, bool& sane)
{
    const int l0 = bitlen_words<LAT_R>(r0), l1 = bitlen_words<LAT_R>(r1);
    const int top = l0 > l1 ? l0 : l1;
    const bool words = __any(active && top > 62 + 31);
    const int scale = top > 62 ? top - 62 : 0;            // x0, x1 are the remainders >> scale
    u64 x0 = leading62<LAT_R>(r0, top, words), x1 = leading62<LAT_R>(r1, top, words);
    // x0 = A*a - B*b, x1 = -C*a + D*b for the original leading parts (a, b); entries only ever grow
    u32 A = 1, B = 0, C = 0, D = 1;
#if C25519_LAT_QUOTIENT_STEPS
    // Euclid on the leading parts with estimated quotients, the two remainders taking turns: x0 -= q x1, then x1 -= q x0.  No
    // role selects (the shift-subtract form below spends most of its 72 instructions per trip on them), and a step removes
    // 1.7 bits on average instead of 1.3.  q comes from a float division of the top 32 bits, scaled so that it can only
    // UNDER-estimate (lat_quot_est); any q keeps the transformation unimodular, a small one just makes less progress.
    auto half_step = [&](u64& xa, const u64 xb, u32& Ma0, u32& Ma1, const u32 Mb0, const u32 Mb1) -> bool {
        u32 q = lat_quot_est(xa, xb);
        const u64 n0 = (u64)Ma0 + (u64)q * Mb0, n1 = (u64)Ma1 + (u64)q * Mb1;
        // keep the matrix below 2^31 and the subtracted leading part above 2^33 (below that its low bits are noise); stop
        // where the exact steps take over (the smaller remainder down to LAT_LEHMER_STOP bits)
        const bool go = active && q != 0 && xb >= ((u64)1 << 33) && n0 < ((u64)1 << 31) && n1 < ((u64)1 << 31)
                        && bitlen64(xb) + scale > LAT_LEHMER_STOP;
        if (go) {
            u64 p = (u64)q * xb;                              // q <= xa / xb: no overflow
            xa -= p;
            Ma0 = (u32)n0; Ma1 = (u32)n1;
        }
        return go;
    };
#pragma unroll 1
    for (int it = 0; it < 40; it++) {
        const bool g0 = half_step(x0, x1, A, B, C, D);
        const bool g1 = half_step(x1, x0, C, D, A, B);
        if (!__any(g0 || g1)) break;
        C25519_LAT_COUNT(lehmer_inner);
    }
#else
#pragma unroll 1
    for (int it = 0; it < 48; it++) {
        const bool c = x0 >= x1;
        const u64 big = c ? x0 : x1, small = c ? x1 : x0;
        const u32 ms0 = c ? C : A, ms1 = c ? D : B, mb0 = c ? A : C, mb1 = c ? B : D;     // rows of small / big
        int k = bitlen64(big) - bitlen64(small);
        u64 sh = small << k;
        if (sh > big) { k -= 1; sh >>= 1; }
        const u64 n0 = (u64)mb0 + ((u64)ms0 << k), n1 = (u64)mb1 + ((u64)ms1 << k);
        // keep the matrix below 2^31 and the smaller leading part above 2^33 (below that its low bits are noise); stop
        // where the exact steps take over (the smaller remainder down to LAT_LEHMER_STOP bits)
        const bool go = active && small >= ((u64)1 << 33) && k < 31 && n0 < ((u64)1 << 31) && n1 < ((u64)1 << 31)
                        && bitlen64(small) + scale > LAT_LEHMER_STOP;
        if (!__any(go)) break;
        C25519_LAT_COUNT(lehmer_inner);
        if (go) {
            const u64 nb = big - sh;
            x0 = c ? nb : x0; x1 = c ? x1 : nb;
            A = c ? (u32)n0 : A; B = c ? (u32)n1 : B;
            C = c ? C : (u32)n0; D = c ? D : (u32)n1;
        }
    }
#endif
    const bool progressed = active && !(A == 1 && B == 0 && C == 0 && D == 1);
    // apply exactly: (r0, r1) <- (A r0 - B r1, -C r0 + D r1), same for the cofactors (two's complement, sign-extended)
    u32 x[LAT_R + 1], y[LAT_R + 1], n0[LAT_R + 1], n1[LAT_R + 1];
#pragma unroll
    for (int i = 0; i < LAT_R; i++) { x[i] = r0[i]; y[i] = r1[i]; }
    x[LAT_R] = y[LAT_R] = 0;
    lin_comb<LAT_R + 1>(n0, x, A, y, B);
    lin_comb<LAT_R + 1>(n1, y, D, x, C);
    u32 tx[LAT_T + 1], ty[LAT_T + 1], m0[LAT_T + 1], m1[LAT_T + 1];
#pragma unroll
    for (int i = 0; i < LAT_T; i++) { tx[i] = t0[i]; ty[i] = t1[i]; }
    tx[LAT_T] = 0u - (t0[LAT_T - 1] >> 31);
    ty[LAT_T] = 0u - (t1[LAT_T - 1] >> 31);
    lin_comb<LAT_T + 1>(m0, tx, A, ty, B);
    lin_comb<LAT_T + 1>(m1, ty, D, tx, C);
    // a remainder that came out negative (the leading-bits quotient overshot): flip the whole vector
    if (n0[LAT_R] >> 31) { negate_words<LAT_R + 1>(n0); negate_words<LAT_T + 1>(m0); }
    if (n1[LAT_R] >> 31) { negate_words<LAT_R + 1>(n1); negate_words<LAT_T + 1>(m1); }
    if (progressed) {
        sane = sane && n0[LAT_R] == 0 && n1[LAT_R] == 0;
#pragma unroll
        for (int i = 0; i < LAT_R; i++) { r0[i] = n0[i]; r1[i] = n1[i]; }
#pragma unroll
        for (int i = 0; i < LAT_T; i++) { t0[i] = m0[i]; t1[i] = m1[i]; }
    }
    return progressed;
}

// h (< 2^256, normally canonical mod L) -> rho (odd, > 0), |tau|, sign of tau, as 5-word little-endian magnitudes.
// Returns all-ones if both fit the walk (bit length <= cap_bits, which is LAT_CAP_BITS in production; a test lowers it
// through the tunable VERIFY_LAT_CAP_BITS to send ordinary signatures down the fall-back), zero otherwise (the caller falls back).
C25519_DEV u32 sc_lattice_short(u32 (&rho)[5], u32 (&tau)[5], u32& tau_negative, const u32 (&h)[8], int cap_bits = LAT_CAP_BITS)
{
    u32 r0[LAT_R], r1[LAT_R], t0[LAT_T] = { 0, 0, 0, 0, 0, 0 }, t1[LAT_T] = { 1, 0, 0, 0, 0, 0 };
    lat_modulus(r0);
#pragma unroll
    for (int i = 0; i < LAT_R; i++) r1[i] = h[i];
    bool sane = true;
    // phase 1: Lehmer steps while both remainders are still well above the target
    {
        bool active = true;
#pragma unroll 1
        for (int guard = 0; guard < 16; guard++) {
            const int l0 = bitlen_words<LAT_R>(r0), l1 = bitlen_words<LAT_R>(r1);
            active = active && sane && (l0 < l1 ? l0 : l1) > LAT_LEHMER_STOP;
            if (!__any(active)) break;
            C25519_LAT_COUNT(lehmer_outer);
            active = lat_lehmer_step(r0, t0, r1, t1, active, sane) && active;
        }
    }
    // phase 2: exact steps.  A step subtracts the (shifted) smaller vector from the larger one; it is wanted while the
    // smaller remainder still has more than 128 bits, and after that only where it improves the vector that will be used.
    bool stopped = false;
#pragma unroll 1
    for (int guard = 0; guard < 400; guard++) {
        const bool c = geq_words<LAT_R>(r0, r1);                       // r0 is the larger one
        const int l0 = bitlen_words<LAT_R>(r0), l1 = bitlen_words<LAT_R>(r1);
        const int lbig = c ? l0 : l1, lsmall = c ? l1 : l0;
        // |T_small| << k has to stay inside the signed 192 bits; if it would not (h with a freak quotient, e.g. h = 1)
        // this lane simply stops: the smaller vector may still be the answer, the larger one is then of no use
        const int k0 = lbig - lsmall;
        u32 mag[LAT_T];
#pragma unroll
        for (int i = 0; i < LAT_T; i++) mag[i] = (c ? t1[i] : t0[i]) ^ (0u - ((c ? t1[LAT_T - 1] : t0[LAT_T - 1]) >> 31));
        const int lt_small = bitlen_words<LAT_T>(mag);
        const bool room = lt_small + k0 <= 32 * LAT_T - 3;
#if C25519_LAT_BALANCED_STOP
        // The first vector whose remainder is below 2^128 has a cofactor below N / 2^128 = 2^127.5 (|T_(i+1)| r_i < N):
        // when that cofactor is odd it is the answer and nothing is left to do.  When it is even, the other vector (odd
        // cofactor) is the answer, and a step on it -- remainder one bit shorter, cofactor up to |T_small| << k -- is wanted
        // only while it shortens the longer of the two: the remainder is the longer one and the new cofactor stays below it.
        // (Round 2's rule ran both remainders down to 128 / 129 bits whatever that did to the cofactor: vectors of up to
        // 139 bits, every second wave starting its walk a digit higher, and four exact steps per wave instead of two.)
        u32 magb[LAT_T];
#pragma unroll
        for (int i = 0; i < LAT_T; i++) magb[i] = (c ? t0[i] : t1[i]) ^ (0u - ((c ? t0[LAT_T - 1] : t1[LAT_T - 1]) >> 31));
        const int lt_big = bitlen_words<LAT_T>(magb);
        const bool small_even = ((c ? t1[0] : t0[0]) & 1u) == 0;
        const bool improve_big = small_even && lbig > lt_big && lt_small + k0 < lbig;
        const bool want = sane && !stopped && room && lsmall != 0 && (lsmall > 128 || improve_big);
#else
        const bool want = sane && !stopped && room && lsmall != 0 && (lsmall > 128 || lbig > 129);
#endif
        stopped = stopped || (sane && !room);
        if (!__any(want)) break;
        C25519_LAT_COUNT(exact_steps);
        u32 big[LAT_R], small[LAT_R], tb[LAT_T], ts[LAT_T], sh[LAT_R], tsh[LAT_T];
#pragma unroll
        for (int i = 0; i < LAT_R; i++) { big[i] = c ? r0[i] : r1[i]; small[i] = c ? r1[i] : r0[i]; }
#pragma unroll
        for (int i = 0; i < LAT_T; i++) { tb[i] = c ? t0[i] : t1[i]; ts[i] = c ? t1[i] : t0[i]; }
        int k = want ? k0 : 0;
        const bool words = __any(k >= 32);
        shl_var<LAT_R>(sh, small, k, words);
        if (!geq_words<LAT_R>(big, sh)) {                              // one bit too far (k >= 1 here)
            k -= 1;
#pragma unroll
            for (int i = 0; i < LAT_R; i++) sh[i] = (sh[i] >> 1) | (i + 1 < LAT_R ? sh[i + 1] << 31 : 0u);
        }
        shl_var<LAT_T>(tsh, ts, k, words);
        u32 borrow = 0, borrow_t = 0;
#pragma unroll
        for (int i = 0; i < LAT_R; i++) {
            const u64 d = (u64)big[i] - sh[i] - borrow;
            big[i] = (u32)d;
            borrow = (u32)(d >> 63);
        }
#pragma unroll
        for (int i = 0; i < LAT_T; i++) {                              // two's complement: T_big -= T_small << k
            const u64 d = (u64)tb[i] - tsh[i] - borrow_t;
            tb[i] = (u32)d;
            borrow_t = (u32)(d >> 63);
        }
        if (want) {
#pragma unroll
            for (int i = 0; i < LAT_R; i++) { r0[i] = c ? big[i] : r0[i]; r1[i] = c ? r1[i] : big[i]; }
#pragma unroll
            for (int i = 0; i < LAT_T; i++) { t0[i] = c ? tb[i] : t0[i]; t1[i] = c ? t1[i] : tb[i]; }
        }
    }
    // the smaller vector if its cofactor is odd, else the other (a basis never has two even cofactors)
    const bool small_is_1 = geq_words<LAT_R>(r0, r1);
    const bool small_odd = ((small_is_1 ? t1[0] : t0[0]) & 1u) != 0;
    const bool use1 = small_is_1 == small_odd;                         // vector 1 is picked
    u32 r[LAT_R], t[LAT_T];
#pragma unroll
    for (int i = 0; i < LAT_R; i++) r[i] = use1 ? r1[i] : r0[i];
#pragma unroll
    for (int i = 0; i < LAT_T; i++) t[i] = use1 ? t1[i] : t0[i];
    const bool t_negative = (t[LAT_T - 1] >> 31) != 0;                 // r = T*h: with rho = |T|, tau = sign(T) * r
    if (t_negative) negate_words<LAT_T>(t);
    tau_negative = t_negative ? 0xffffffffu : 0u;
#pragma unroll
    for (int i = 0; i < 5; i++) { rho[i] = t[i]; tau[i] = r[i]; }
    const bool fits = sane && bitlen_words<LAT_R>(r) <= cap_bits && bitlen_words<LAT_T>(t) <= cap_bits && (t[0] & 1u);
    return fits ? 0xffffffffu : 0u;
}

// sigma = rho * s mod L (canonical); s is the signature's raw 256-bit S
C25519_DEV void sc_mul_short(u32 (&sigma)[8], const u32 (&rho)[5], const u32 (&s)[8])
{
    u32 r8[8];
#pragma unroll
    for (int i = 0; i < 8; i++) r8[i] = i < 5 ? rho[i] : 0u;
    sc_mul(sigma, s, r8);
    sc_mod(sigma);
}

// ---- signed radix-16 digits ----------------------------------------------------------------------------------------------
constexpr int WALK_DIGITS = 40;             // 40 digits in [-8, 7] cover magnitudes below 2^158 (five 32-bit words of nibbles)
// k + 0x888...8 (40 nibbles): nibble i of the sum, minus 8, is the signed digit d_i in [-8, 7] with k = sum d_i 16^i,
// so digits can be read most-significant first without a carry chain.  k < 2^158: the sum stays below 2^160.
C25519_DEV void bias_signed16(u32 (&kb)[5], const u32 (&k)[5])
{
    u64 c = 0;
#pragma unroll
    for (int i = 0; i < 5; i++) {
        c += (u64)k[i] + 0x88888888u;
        kb[i] = (u32)c;
        c >>= 32;
    }
}
// digit `nib` (0..7) of one word of a biased scalar: |d| in 0..8 and all-ones if d < 0
C25519_DEV u32 signed16_of(u32& negative, u32 word, int nib)
{
    const u32 v = (word >> (4 * nib)) & 15u;
    negative = v < 8u ? 0xffffffffu : 0u;
    return v < 8u ? 8u - v : v - 8u;
}

// index of the most significant nonzero signed digit of either biased scalar (0 if there is none): nibbles that still
// equal the bias' 8 are zero digits
C25519_DEV int walk_top_digit(const u32 (&tau_b)[5], const u32 (&rho_b)[5])
{
    int top = 0;
#pragma unroll
    for (int w = 0; w < 5; w++) {
        const u32 v = (tau_b[w] ^ 0x88888888u) | (rho_b[w] ^ 0x88888888u);
        top = v ? 8 * w + ((31 - __builtin_clz(v)) >> 2) : top;
    }
    return top;
}

// sigma recoded for the walk's signed comb (SC_TEETH teeth, SC_COLS bits apart; the recoding of ge_base_mult, ge25519.cuh):
// sigma is made odd (+ L when even: L*B = O) and written with SC_TEETH * SC_COLS digits +-1, w = (sigma' >> 1) | top bit;
// column c = the SC_TEETH bits of w at c, c + SC_COLS, ...: the top one is the column's sign, the others the table row
// (complemented for a negative column).  Stored 16 bits per column in the order the walk consumes them: digit round i
// (four doublings) reads words 2i and 2i + 1 = columns 4i+3, 4i+2 | 4i+1, 4i, the first of them in the low half.  The walk
// reads two words per round instead of holding all of sigma.  sigma < L.
C25519_DEV void sc_comb_columns(u32 (&cols)[SIGMA_WORDS], const u32 (&k)[8])
{
    const u32 even = (k[0] & 1u) - 1u;                       // all-ones when sigma is even
    u32 t[9], w[9];
    u64 c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        c += (u64)k[i] + (K_L[i] & even);
        t[i] = (u32)c;
        c >>= 32;
    }
    t[8] = (u32)c;                                           // 0: sigma + L < 2^254
#pragma unroll
    for (int i = 0; i < 8; i++) w[i] = (t[i] >> 1) | (t[i + 1] << 31);
    w[8] = 0;
    constexpr int TOP = SC_TEETH * SC_COLS - 1;              // digit TOP is +1
    w[TOP >> 5] |= 1u << (TOP & 31);
    auto column = [&](int col) -> u32 {
        u32 idx = 0;
        if (col >= SC_COLS) return 0u;
#pragma unroll
        for (int j = 0; j < SC_TEETH; j++) {
            const int bit = SC_COLS * j + col;
            idx |= ((w[bit >> 5] >> (bit & 31)) & 1u) << j;
        }
        return idx;
    };
#pragma unroll
    for (int r = 0; r < SC_ROUNDS; r++) {
        cols[2 * r] = column(4 * r + 3) | (column(4 * r + 2) << 16);
        cols[2 * r + 1] = column(4 * r + 1) | (column(4 * r) << 16);
    }
}

// ---- packed table rows -------------------------------------------------------------------------------------------------
// A precomputed point (Y+X, Y-X, 2dT, 2Z) is stored as four 256-bit integers, 8 words each: ONE 128-byte row, 128-byte
// aligned, instead of 40 limbs (160 bytes over two cache lines).  Measured before it was built (timing experiment,
// profiles/r03_ab_verify_structure.txt block 7): one line and 8 loads per row instead of two lines and 12 loads is worth 6 %
// of the verification pass -- the texture path's share of a VALU-bound kernel's clock -- against ~18 unpack instructions
// per field.  Packing is a plain positional sum (limb i at bit ceil(25.5 i)), so limbs need not be strictly below 2^w, only
// the value below 2^256 (true of a product's output and of fe_carry32's); unpacking is fe_from_words, bit 255 included.
constexpr int ROW_WORDS = 32;
#ifndef C25519_WALK_PREFETCH
#define C25519_WALK_PREFETCH 1       // A/B switch: 0 = the walk loads each row field right before the product that uses it
#endif

C25519_DEV void fe_pack_words(u32 (&w)[8], const fe& a)
{
    u64 acc = (u64)a.v[0] + ((u64)a.v[1] << 26);
    w[0] = (u32)acc; acc >>= 32;
    acc += (u64)a.v[2] << 19;
    w[1] = (u32)acc; acc >>= 32;
    acc += (u64)a.v[3] << 13;
    w[2] = (u32)acc; acc >>= 32;
    acc += (u64)a.v[4] << 6;
    w[3] = (u32)acc; acc >>= 32;
    acc += (u64)a.v[5] + ((u64)a.v[6] << 25);
    w[4] = (u32)acc; acc >>= 32;
    acc += (u64)a.v[7] << 19;
    w[5] = (u32)acc; acc >>= 32;
    acc += (u64)a.v[8] << 12;
    w[6] = (u32)acc; acc >>= 32;
    acc += (u64)a.v[9] << 6;
    w[7] = (u32)acc;
}
// one field of a row: p = row + 8 * field (32-byte aligned)
C25519_DEV void load_field(fe& f, const u32* p)
{
    const uint4* q = reinterpret_cast<const uint4*>(p);
    const uint4 a = q[0], b = q[1];
    const u32 w[8] = { a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w };
    fe_from_words(f, w);
}
C25519_DEV void store_field(u32* p, const fe& f)
{
    u32 w[8];
    fe_pack_words(w, f);
    uint4* q = reinterpret_cast<uint4*>(p);
    q[0] = make_uint4(w[0], w[1], w[2], w[3]);
    q[1] = make_uint4(w[4], w[5], w[6], w[7]);
}
C25519_DEV void row_load_pe(ge_pe& q, const u32* row)
{
    load_field(q.ypx, row);
    load_field(q.ymx, row + 8);
    load_field(q.t2d, row + 16);
    load_field(q.z2, row + 24);
}
C25519_DEV void row_store_neutral(u32* row)             // (Y+X, Y-X, 2dT, 2Z) of the neutral element: 1, 1, 0, 2
{
    uint4* q = reinterpret_cast<uint4*>(row);
    const uint4 zero = make_uint4(0, 0, 0, 0);
    q[0] = make_uint4(1, 0, 0, 0); q[1] = zero;
    q[2] = make_uint4(1, 0, 0, 0); q[3] = zero;
    q[4] = zero;                   q[5] = zero;
    q[6] = make_uint4(2, 0, 0, 0); q[7] = zero;
}

// ---- per-lane window tables: rows 0..8 = 0, P, 2P, ..., 8P as packed rows ------------------------------------------------------
constexpr int WTABLE_ROWS = 9;
constexpr size_t WTABLE_WORDS = WTABLE_ROWS * ROW_WORDS;  // 288 words = 1152 bytes per table

// (x, y) affine.  ONE extended point lives in registers: P's own row is re-read from the table for the three "+ P"
// steps (ge_add_pe_row below), 2P and 4P for the doublings that restart from them (a product and two carry passes each):
// P, 2P, 3P, 6P, 7P, then (2P ->) 4P, 5P, then (4P ->) 8P.  Built this way the tables fit the walk kernel's registers.
C25519_DEV void wtable_build(u32* rows, const fe& x, const fe& y);

// q <- -q when neg is all-ones: swap Y+X and Y-X, negate 2dT
C25519_DEV void pe_cond_neg(ge_pe& q, u32 neg)
{
    fe t, n;
    fe_select(t, neg, q.ymx, q.ypx);
    fe_select(q.ymx, neg, q.ypx, q.ymx);
    q.ypx = t;
    fe_neg(n, q.t2d);                          // 2p - t2d: beta 2, accepted by ge_add_pe's products
    fe_select(q.t2d, neg, n, q.t2d);
}

// S += (neg ? -row : row), the row read from memory one field at a time, each right before the product that consumes it:
// a whole row in registers (40) on top of the accumulator (40) and the addition's temporaries is what pushes the walk
// over the register budget.  Negation is free here: -q swaps Y+X and Y-X (two field offsets) and negates 2dT (ten
// subtractions), instead of thirty selects on a loaded row.  The sums B-A, B+A are formed as soon as A and B exist and
// D-C, D+C as soon as C and D do, so at most four temporaries live beside the accumulator.
template <bool NEED_T>
C25519_DEV void ge_add_pe_row(ge_ext& S, const u32* row, u32 neg)
{
    const u32* p_ypx = row + (neg ? 8 : 0);         // field that multiplies Y+X
    const u32* p_ymx = row + (neg ? 0 : 8);         // field that multiplies Y-X
    fe q, a, b, e, f, g, h;
    fe_sub(a, S.Y, S.X);
    load_field(q, p_ymx);
    fe_mul(a, a, q);
    fe_add(b, S.Y, S.X);
    load_field(q, p_ypx);
    fe_mul(b, b, q);
    fe_sub(e, b, a);
    fe_add(h, b, a);
    load_field(q, row + 16);
    fe_neg(a, q);                                    // 2p - t2d: beta 2, fine as the second operand of a product
    fe_select(q, neg, a, q);
    fe_mul(a, S.T, q);                               // C
    load_field(q, row + 24);
    fe_mul(b, S.Z, q);                               // D
    fe_sub(f, b, a);                                 // beta 3: still a legal second operand
    fe_add(g, b, a);
    // second operands f, h, f, h: their 19-multiples (nine v_mul_lo_u32 each) are formed twice, not four times, and the
    // doubled odd limbs of the first operands e, e, g, g likewise
    fe_mul(S.X, e, f);
    if (NEED_T) fe_mul(S.T, e, h);
    fe_mul(S.Z, g, f);
    fe_mul(S.Y, g, h);
}

// The same addition from a row already fetched into registers (its 32 packed words).  The walk fetches the two rows of a
// digit round at the TOP of the round -- their addresses only depend on the round's digits -- so the loads are 12 000
// cycles old when the additions want them: measured 4.5 % of the pass against loading each field right before its product
// (profiles/r03_ab_verify_structure.txt block 8), at two waves per SIMD instead of three (64 more registers), which the
// VALU-bound walk does not mind.  -row: the words of Y+X and Y-X trade places before they are unpacked.
struct packed_row { uint4 q[8]; };
C25519_DEV void row_fetch(packed_row& r, const u32* row)
{
    const uint4* p = reinterpret_cast<const uint4*>(row);
#pragma unroll
    for (int i = 0; i < 8; i++) r.q[i] = p[i];
}
// field k of the row, or field k_neg where neg is all-ones
C25519_DEV void field_of(fe& f, const packed_row& r, int k, int k_neg, u32 neg)
{
    const uint4 a = r.q[2 * k], b = r.q[2 * k + 1], c = r.q[2 * k_neg], d = r.q[2 * k_neg + 1];
    const u32 wa[8] = { a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w }, wb[8] = { c.x, c.y, c.z, c.w, d.x, d.y, d.z, d.w };
    u32 w[8];
#pragma unroll
    for (int i = 0; i < 8; i++) w[i] = neg ? wb[i] : wa[i];
    fe_from_words(f, w);
}
template <bool NEED_T>
C25519_DEV void ge_add_pe_regs(ge_ext& S, const packed_row& r, u32 neg)
{
    fe q, a, b, e, f, g, h;
    fe_sub(a, S.Y, S.X);
    field_of(q, r, 1, 0, neg);
    fe_mul(a, a, q);
    fe_add(b, S.Y, S.X);
    field_of(q, r, 0, 1, neg);
    fe_mul(b, b, q);
    fe_sub(e, b, a);
    fe_add(h, b, a);
    field_of(q, r, 2, 2, 0u);
    fe_neg(a, q);                                    // 2p - t2d: beta 2, fine as the second operand of a product
    fe_select(q, neg, a, q);
    fe_mul(a, S.T, q);                               // C
    field_of(q, r, 3, 3, 0u);
    fe_mul(b, S.Z, q);                               // D
    fe_sub(f, b, a);
    fe_add(g, b, a);
    fe_mul(S.X, e, f);
    if (NEED_T) fe_mul(S.T, e, h);
    fe_mul(S.Z, g, f);
    fe_mul(S.Y, g, h);
}

// S += row r of the limb-major LDS base table ([30][256] words), one field at a time like ge_add_pe_row.  need_t is a
// run-time (wave-uniform) flag on purpose: ONE copy of the addition in the walk's loop saves 28 registers against the two
// template instances of ge_add_pa (154 against 182 when this was measured, without the row prefetch).
C25519_DEV void ge_add_pa_lds(ge_ext& S, const u32* tbl, u32 r, bool need_t)
{
    fe q, a, b, e, f, g, h;
    fe_sub(a, S.Y, S.X);
#pragma unroll
    for (int i = 0; i < 10; i++) q.v[i] = tbl[(10 + i) * 256 + r];
    fe_mul(a, a, q);
    fe_add(b, S.Y, S.X);
#pragma unroll
    for (int i = 0; i < 10; i++) q.v[i] = tbl[i * 256 + r];
    fe_mul(b, b, q);
    fe_sub(e, b, a);
    fe_add(h, b, a);
#pragma unroll
    for (int i = 0; i < 10; i++) q.v[i] = tbl[(20 + i) * 256 + r];
    fe_mul(a, S.T, q);                               // C
    fe_add(b, S.Z, S.Z);                             // D
    fe_sub(f, b, a);
    fe_add(g, b, a);
    fe_mul(S.X, f, e);
    if (need_t) fe_mul(S.T, e, h);
    fe_mul(S.Z, f, g);
    fe_mul(S.Y, g, h);
}

// S += column c of the walk's signed comb table (limb-major LDS table [30][SC_ROWS] words): bit SC_TEETH-1 of c is the sign
// (0: the column is negative -- row ~c, negated on the way out of LDS: (y+x, y-x, 2dxy) -> (y-x, y+x, -2dxy)), one field at
// a time like ge_add_pa_lds, need_t a run-time flag for the same reason.
C25519_DEV void ge_add_pa_comb(ge_ext& S, const u32* tbl, u32 c, bool need_t)
{
    const u32 neg = ((c >> (SC_TEETH - 1)) & 1u) - 1u;       // all-ones: negative column
    const u32 r = (c ^ neg) & (u32)(SC_ROWS - 1);
    const u32* p_ypx = tbl + (neg ? 10 * SC_ROWS : 0) + r;
    const u32* p_ymx = tbl + (neg ? 0 : 10 * SC_ROWS) + r;
    fe q, a, b, e, f, g, h;
    fe_sub(a, S.Y, S.X);
#pragma unroll
    for (int i = 0; i < 10; i++) q.v[i] = p_ymx[i * SC_ROWS];
    fe_mul(a, a, q);
    fe_add(b, S.Y, S.X);
#pragma unroll
    for (int i = 0; i < 10; i++) q.v[i] = p_ypx[i * SC_ROWS];
    fe_mul(b, b, q);
    fe_sub(e, b, a);
    fe_add(h, b, a);
#pragma unroll
    for (int i = 0; i < 10; i++) q.v[i] = tbl[(20 + i) * SC_ROWS + r];
    fe_neg(a, q);                                    // 2p - 2dxy: beta 2, fine as the second operand of a product
    fe_select(q, neg, a, q);
    fe_mul(a, S.T, q);                               // C
    fe_add(b, S.Z, S.Z);                             // D
    fe_sub(f, b, a);
    fe_add(g, b, a);
    fe_mul(S.X, f, e);
    if (need_t) fe_mul(S.T, e, h);
    fe_mul(S.Z, f, g);
    fe_mul(S.Y, g, h);
}

// row <- precomputed form of S (ge_to_pe), packed, a field at a time
C25519_DEV void ge_store_pe_row(u32* row, const ge_ext& S)
{
    fe t, a;
    fe_add(t, S.Y, S.X);  fe_carry32(a, t);  store_field(row, a);
    fe_sub(t, S.Y, S.X);  fe_carry32(a, t);  store_field(row + 8, a);
    fe_mul(a, S.T, fe_const(K_2D));          store_field(row + 16, a);
    fe_add(t, S.Z, S.Z);  fe_carry32(a, t);  store_field(row + 24, a);
}

C25519_DEV void wtable_build(u32* rows, const fe& x, const fe& y)
{
    ge_ext S;
    row_store_neutral(rows);
    S.X = x; S.Y = y;
    fe_mul(S.T, x, y);
    fe_set_u32(S.Z, 1);
    ge_store_pe_row(rows + ROW_WORDS, S);
    // seven steps, alternately a doubling and "+ P", as a loop (one copy of each operation, and the scheduler cannot
    // stretch live ranges across steps: the straight-line version took 207 registers, this one fits the 168 of three
    // waves per SIMD): rows 2, 3, 6, 7, then from 2P: 4, 5, then from 4P: 8
#pragma unroll 1
    for (int step = 0; step < 7; step++) {
        if (step == 4 || step == 6) {
            ge_pe pe;
            row_load_pe(pe, rows + (step == 4 ? 2 : 4) * ROW_WORDS);
            ge_from_pe(S, pe);
        }
        if (step & 1) ge_add_pe_row<true>(S, rows + ROW_WORDS, 0u);
        else ge_double<true>(S);
        ge_store_pe_row(rows + ((0x8547632u >> (4 * step)) & 15u) * ROW_WORDS, S);
    }
}

// ---- the reference's 16-row 4-fold table, built with the streamed operations above -----------------------------------------------
// qtable_build (ge25519.cuh; ed25519_Verify_Init :199-229) for a lane-private limb table: the same points in the same
// order by the same formulas -- row top+s = Q + row s with Q the extended and row s the precomputed operand, as the
// reference's edp_AddPoint has them (that matters: for an off-curve "key" the values depend on it) -- but each sum is
// formed in ONE extended point from a row streamed out of memory and leaves through ge_store_pe_row, so the build needs the
// registers of the walk (Q, the sum, a product's temporaries), not 175.  rows: 16 packed rows.  Q is consumed.
C25519_DEV void qtable_build_streamed(u32* rows, ge_ext& Q)
{
    row_store_neutral(rows);
    ge_store_pe_row(rows + ROW_WORDS, Q);
#pragma unroll 1
    for (int blk = 1; blk < 4; blk++) {               // Q <- 2^64 Q, then fill rows [2^blk, 2^(blk+1))
#pragma unroll 1
        for (int i = 0; i < 63; i++) ge_double<false>(Q);
        ge_double<true>(Q);
        const int top = 1 << blk;
        ge_store_pe_row(rows + top * ROW_WORDS, Q);
        // the sums restart from the row just written (precomputed -> extended is one product; the coordinates come back
        // doubled, which the homogeneous formulas do not see), so Q itself does not have to stay in registers beside them
#pragma unroll 1
        for (int s = 1; s <= top; s++) {
            ge_pe pe;
            row_load_pe(pe, rows + top * ROW_WORDS);
            ge_from_pe(Q, pe);
            if (s == top) break;                      // ... and Q is back for the next block's doublings
            ge_add_pe_row<true>(Q, rows + s * ROW_WORDS, 0u);
            ge_store_pe_row(rows + (top + s) * ROW_WORDS, Q);
        }
    }
}

// S = s*B + h*Q by the interleaved 4-fold / 8-fold walk (ge_poly_mult, ge25519.cuh; edp_PolyPointMultiply :243-280) with the
// rows of the lane's 4-fold table and of the LDS base table streamed into the additions.  s and h are consumed.
C25519_DEV void ge_poly_mult_streamed(ge_ext& S, u32 (&s)[8], u32 (&h)[8], const u32* rows, const u32* lds_tbl)
{
    {
        ge_pe pe;
        row_load_pe(pe, rows + fold4_next(h, false) * ROW_WORDS);
        ge_from_pe(S, pe);
    }
#pragma unroll 1
    for (int i = 1; i < 32; i++) {
        ge_double(S);
        ge_add_pe_row<false>(S, rows + fold4_next(h, false) * ROW_WORDS, 0u);
    }
#pragma unroll 1
    for (int i = 32; i < 64; i++) {
        ge_double(S);
        ge_add_pa_lds(S, lds_tbl, fold8_next(s), true);     // T feeds the addition that follows
        ge_add_pe_row<false>(S, rows + fold4_next(h, true) * ROW_WORDS, 0u);
    }
}

// the reference-order path for one element, start to finish (own table, own inversion): what ed25519_VerifySignature does
// (ed25519_verify.c:163-176 = Verify_Init + Verify_Check), on the streamed forms above: 180 registers instead of the 268 of
// the generic ones, two waves per SIMD.  lane_table: 16 packed rows of lane-private, 128-byte aligned memory.  Sw is consumed.
// enc_out (tests): enc(T), what the verdict compares with the R bytes.
C25519_DEV int ed_verify_reference_order(const u32 (&pkw)[8], const u32 (&Rw)[8], u32 (&Sw)[8], const uint8_t* msg, size_t len,
                                         u32* lane_table, const u32* lds_tbl, u32* enc_out = nullptr)
{
    u32 h[8], enc[8], xw[8], yw[8];
    {
        ge_ext Q;
        ed_decode_neg_key(Q, pkw);
        qtable_build_streamed(lane_table, Q);
    }
    ed_hram(h, Rw, pkw, msg, len);
    sc_mod(h);
    ge_ext T;
    ge_poly_mult_streamed(T, Sw, h, lane_table, lds_tbl);
    ge_to_affine_words(xw, yw, T);                            // z^(p-2): Z == 0 gives 0 like the reference
    ge_pack(enc, xw, yw);
    u32 diff = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) diff |= enc[j] ^ Rw[j];
    if (enc_out) {
#pragma unroll
        for (int j = 0; j < 8; j++) enc_out[j] = enc[j];
    }
    return diff == 0 ? 1 : 0;
}

// ---- the walk ----------------------------------------------------------------------------------------------------------
// Where a lane's scalars live while it walks: struct-of-arrays scratch written by the scalars kernel (word w of element i
// at base[w * n + i]; the host emulation passes n = 1, i = 0).  The walk fetches ONE word of each per digit round -- the
// round's nibble of tau and rho, the round's four comb columns of sigma -- instead of holding 18 words in registers.
struct WalkScalars {
    const u32 *sigma_cols, *tau, *rho;
    size_t n, i;
    C25519_DEV u32 tau_word(int w) const { return tau[(size_t)w * n + i]; }
    C25519_DEV u32 rho_word(int w) const { return rho[(size_t)w * n + i]; }
    C25519_DEV u32 sigma_word(int w) const { return sigma_cols[(size_t)w * n + i]; }
};

// W = sigma*B + tau*Q + rho*Rn from the two window tables (Q and Rn = -R already carry the signs of tau and of the
// equation), the biased scalars and the LDS base table; returns all-ones iff W is the neutral element.
// `top`: the walk starts at this digit; every digit above it must be zero in both scalars, and top >= SC_ROUNDS (sigma's
// columns ride the last SC_ROUNDS digit rounds; the kernels pass at least 8).  The kernels pass the maximum of walk_top_digit() over the wave: typical short vectors
// have 127-131 bits, so a wave starts at digit 32 or 33 and walks 33-34 rounds of four doublings and two table
// additions; leading zero digits would have added the neutral row, so skipping them is exact.
// tq, tr: the element's two window tables (WTABLE_ROWS packed rows each).
C25519_DEV u32 ge_walk_is_neutral(const WalkScalars& sc, const u32* tq, const u32* tr, const u32* lds_tbl, int top)
{
    ge_ext S;
    {
        ge_pe pe;
        u32 neg;
        const u32 m = signed16_of(neg, sc.tau_word(top >> 3), top & 7);
        row_load_pe(pe, tq + m * ROW_WORDS);
        pe_cond_neg(pe, neg);
        ge_from_pe(S, pe);
        const u32 m2 = signed16_of(neg, sc.rho_word(top >> 3), top & 7);
        ge_add_pe_row<false>(S, tr + m2 * ROW_WORDS, neg);
    }
    // sigma's comb columns ride on the last SC_COLS doublings (the reference's own trick with its 8-fold table,
    // ed25519_verify.c:266-279): column c is added with c doublings to go
#pragma unroll 1
    for (int i = top - 1; i >= 0; i--) {
        const u32 tw = sc.tau_word(i >> 3), rw = sc.rho_word(i >> 3);      // in flight under the doublings
#if C25519_WALK_PREFETCH
        u32 negq, negr;
        packed_row rq, rr;
        row_fetch(rq, tq + signed16_of(negq, tw, i & 7) * ROW_WORDS);
        row_fetch(rr, tr + signed16_of(negr, rw, i & 7) * ROW_WORDS);
#endif
        if (i >= SC_ROUNDS) {
#pragma unroll 1
            for (int j = 0; j < 3; j++) ge_double<false>(S);
            ge_double<true>(S);
        } else {
            u64 cols = (u64)sc.sigma_word(2 * i) | ((u64)sc.sigma_word(2 * i + 1) << 32);
#pragma unroll 1
            for (int j = 0; j < 4; j++) {
                ge_double<true>(S);
                if (4 * i + 3 - j < SC_COLS)                         // wave-uniform: the first round may carry fewer than four
                    ge_add_pa_comb(S, lds_tbl, (u32)cols & 0xffffu, j == 3);   // T feeds the key-table addition behind the last one
                cols >>= 16;
            }
        }
#if C25519_WALK_PREFETCH
        ge_add_pe_regs<true>(S, rq, negq);
        ge_add_pe_regs<false>(S, rr, negr);
#else
        u32 neg;
        const u32 mq = signed16_of(neg, tw, i & 7);
        ge_add_pe_row<true>(S, tq + mq * ROW_WORDS, neg);
        const u32 mr = signed16_of(neg, rw, i & 7);
        ge_add_pe_row<false>(S, tr + mr * ROW_WORDS, neg);
#endif
    }
    // neutral element: X == 0 and Y == Z (Z != 0 for on-curve inputs under the complete law)
    u32 xw[8], dw[8], acc = 0;
    fe d;
    fe_sub(d, S.Y, S.Z);
    fe_to_words(xw, S.X);
    fe_to_words(dw, d);
#pragma unroll
    for (int i = 0; i < 8; i++) acc |= xw[i] | dw[i];
    return acc == 0 ? 0xffffffffu : 0u;
}

// ---- one element, in four steps (four kernels in engine.hip; the CPU tests chain them) --------------------------------------
// step 1, integers only: h = H(R || pk || m) mod L, the short vector, sigma = rho * s mod L.  rho and tau come back
// BIASED (bias_signed16) and sigma as its signed comb columns (sc_comb_columns), ready for the walk.  Returns all-ones if
// the vector fits the walk.
C25519_DEV u32 ed_verify_fast_scalars(u32 (&sigma_cols)[SIGMA_WORDS], u32 (&rho)[5], u32 (&tau)[5], u32& tau_negative, const u32 (&pkw)[8],
                                      const u32 (&Rw)[8], const u32 (&Sw)[8], const uint8_t* msg, size_t len, int cap_bits = LAT_CAP_BITS)
{
    u32 h[8], sigma[8];
    {
        u32 le[16];
        u64 pre[8], dg[8];
        sha512_words_from_le32(pre, Rw);
        sha512_words_from_le32(pre + 4, pkw);
        sha512_prefixed<8>(dg, pre, msg, len);
        sha512_digest_le_words(le, dg);
        sc_reduce512(h, le);
        sc_mod(h);
    }
    const u32 lat_ok = sc_lattice_short(rho, tau, tau_negative, h, cap_bits);
    sc_mul_short(sigma, rho, Sw);
    sc_comb_columns(sigma_cols, sigma);
    u32 b[5];
    bias_signed16(b, rho);
#pragma unroll
    for (int i = 0; i < 5; i++) rho[i] = b[i];
    bias_signed16(b, tau);
#pragma unroll
    for (int i = 0; i < 5; i++) tau[i] = b[i];
    return lat_ok;
}

// step 2, one point per call (the kernel gives the key and R of an element to two different lanes): y from the 32 bytes
// with bit 255 stripped, x with the requested parity.
//   is_r = 0:       the key A, decoded as -A exactly as ed25519_Verify_Init does (inverted parity, :191-197), and negated
//                   again when tau < 0 (the walk then uses |tau| on -Q);
//   is_r = all-ones: R, which must be the canonical encoding of a curve point (y < p and the sign bit an encoder would
//                   have produced; otherwise the reference's byte comparison cannot succeed for an on-curve key),
//                   negated: the walk adds rho * (-R).
// Returns all-ones iff the point is on the curve (and, for R, canonically encoded).
C25519_DEV u32 ed_verify_fast_decode(fe& X, fe& Y, const u32 (&w)[8], u32 is_r, u32 tau_negative)
{
    u32 yw[8], cw[8];
#pragma unroll
    for (int i = 0; i < 8; i++) yw[i] = w[i];
    const u32 sign = yw[7] >> 31;
    yw[7] &= 0x7fffffffu;
    fe_from_words(Y, yw);
    const u32 parity = sign ^ (~is_r & 1u);
    u32 ok = ge_calc_x_checked(X, Y, parity);
    fe_to_words(cw, Y);
    u32 diff = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) diff |= cw[i] ^ yw[i];
    fe_to_words(cw, X);
    const u32 canonical = (diff == 0 && ((cw[0] ^ parity) & 1u) == 0) ? 0xffffffffu : 0u;
    ok &= ~is_r | canonical;
    fe t;
    fe_neg(t, X);
    fe_carry32(t, t);
    fe_select(X, is_r | tau_negative, t, X);
    return ok;
}

// step 3: wtable_build above, once per point (the kernel gives the two tables of an element to two different lanes).
// step 4: ge_walk_is_neutral above.

}  // namespace c25519
