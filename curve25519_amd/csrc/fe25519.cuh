// curve25519_amd/csrc/fe25519.cuh -- GF(2^255-19) for gfx950, one field element per lane.
//
// Device replacement for the reference's L0 layer (source/curve25519_mehdi.c: ecp_MulReduce :278,
// ecp_SqrReduce :310, ecp_AddReduce :134, ecp_SubReduce :161, ecp_WordMulAddReduce :243, ecp_Mod :185,
// ecp_Inverse :340; source/ed25519_verify.c: ecp_ModExp2523 :116).  Only the bytes that leave a kernel
// have to match the reference (they are canonical there too), so the in-register form is free.
//
// Representation: 5 x 51-bit limbs, each held as a (26-bit, 25-bit) pair of 32-bit VGPRs, i.e. ten
// unsaturated limbs of radix 2^25.5.  Why not saturated 4x64 / 8x32: measured on MI355X
// (profiles/r01_valu_rates.txt) v_mad_u64_u32 issues at ~30 T lane-op/s, barely slower than
// v_addc_co_u32 (~35 T/s), so a saturated schoolbook product pays one carry instruction per multiply
// AND serialises every step on VCC.  With 2^25.5 limbs a product is a pure chain of v_mad_u64_u32 into
// 64-bit accumulators (no carries, no VCC), the x19 fold of 2^255 = 19 is absorbed by pre-scaling one
// operand, and add/sub are ten plain v_add_u32 / v_sub_u32.  The carry out of column k is the addend the
// first MAD of column k+1 starts from, so carry propagation costs a 64-bit shift and a mask per limb.
//
// Bound contract (beta = limb / 2^w, w = 26 for even limbs, 25 for odd):
//   reduced          : output of mul / sqr / mul121665_add / from_bytes, limb < 2^w + 2^17
//   fe_add(a,b)      : beta_a + beta_b
//   fe_sub(a,b)      : beta_a + 2   (adds 2p limb-wise; needs b reduced)
//   fe_mul(a,b)      : needs beta_a <= 5, beta_b <= 3.3 (19*b_j and 2*a_i must fit 32 bits and every
//                      column sum must stay below 2^64: 124.5 * beta_a * beta_b * 2^52 < 2^64)
//   fe_sqr(a)        : needs beta_a <= 3.3
// tools/fe_bounds.py replays every formula in the kernels against this contract.
#pragma once
// The handful of gfx950 instruction sequences the field layer is built from (v_mad_u64_u32 column chains,
// v_add_u32 doubling) live in valu_gfx950.cuh.  The CPU unit tests of THIS source (tests/host_emul/) pre-include
// a C model of exactly those primitives and define C25519_VALU_PRIMITIVES; the product build never does.
#ifndef C25519_VALU_PRIMITIVES
#include "valu_gfx950.cuh"
#endif
#include "safegcd25519.cuh"

namespace c25519 {

constexpr u32 M26 = 0x3ffffffu;
constexpr u32 M25 = 0x1ffffffu;

struct fe { u32 v[10]; };

// A/B knob: the limb mask behind each product column at low wave priority too (valu_gfx950.cuh: C25519_VOP2_RUN_*)
// every field addition / subtraction / negation / select is such a run (A/B knob C25519_FE_PRIO, default on); the operand
// doublings at the head of a product too (C25519_MASK_PRIO >= 1)
#ifndef C25519_FE_PRIO
#define C25519_FE_PRIO 1
#endif
#if C25519_FE_PRIO
#define C25519_FE_RUN_BEGIN() C25519_VOP2_RUN_BEGIN()
#define C25519_FE_RUN_END() C25519_VOP2_RUN_END()
#else
#define C25519_FE_RUN_BEGIN() do { } while (0)
#define C25519_FE_RUN_END() do { } while (0)
#endif
#ifndef C25519_MASK_PRIO
#define C25519_MASK_PRIO 1
#endif
#if defined(C25519_MASK_PRIO) && C25519_MASK_PRIO
#define C25519_MASK_RUN_BEGIN() C25519_VOP2_RUN_BEGIN()
#define C25519_MASK_RUN_END() C25519_VOP2_RUN_END()
#else
#define C25519_MASK_RUN_BEGIN() do { } while (0)
#define C25519_MASK_RUN_END() do { } while (0)
#endif

C25519_DEV constexpr int fe_w(int i) { return (i & 1) ? 25 : 26; }
C25519_DEV constexpr u32 fe_mask(int i) { return (i & 1) ? M25 : M26; }
// limbs of 2p: every limb stays >= 0 after subtracting a reduced element
C25519_DEV constexpr u32 fe_2p(int i) { return i == 0 ? 0x7ffffdau : ((i & 1) ? 0x3fffffeu : 0x7fffffeu); }

C25519_DEV void fe_set_u32(fe& r, u32 x)
{
#pragma unroll
    for (int i = 1; i < 10; i++) r.v[i] = 0;
    r.v[0] = x;
}

C25519_DEV void fe_add(fe& r, const fe& a, const fe& b)
{
    C25519_FE_RUN_BEGIN();
#pragma unroll
    for (int i = 0; i < 10; i++) r.v[i] = a.v[i] + b.v[i];
    C25519_FE_RUN_END();
}

// r = a - b + 2p  (b must be reduced so that no limb goes negative)
C25519_DEV void fe_sub(fe& r, const fe& a, const fe& b)
{
    C25519_FE_RUN_BEGIN();
#pragma unroll
    for (int i = 0; i < 10; i++) r.v[i] = a.v[i] + fe_2p(i) - b.v[i];
    C25519_FE_RUN_END();
}

// r = 2p - a   (the reference negates with _w_maxP - A, ed25519_sign.c:130)
C25519_DEV void fe_neg(fe& r, const fe& a)
{
    C25519_FE_RUN_BEGIN();
#pragma unroll
    for (int i = 0; i < 10; i++) r.v[i] = fe_2p(i) - a.v[i];
    C25519_FE_RUN_END();
}

// branch-free select: r = mask ? a : b, mask is all-ones or zero
C25519_DEV void fe_select(fe& r, u32 mask, const fe& a, const fe& b)
{
    C25519_FE_RUN_BEGIN();
#pragma unroll
    for (int i = 0; i < 10; i++) r.v[i] = (a.v[i] & mask) | (b.v[i] & ~mask);
    C25519_FE_RUN_END();
}

// limbs l[0..9] hold the masked columns, `carry` is what left column 9: fold it back times 19
C25519_DEV void fe_finish_chain(fe& r, u32 (&l)[10], u64 carry)
{
#ifdef C25519_FENCE_FIELD                 // A/B knob (profiles/r03_ab_fence.txt): a fence behind every product
    C25519_SCHED_FENCE();
#endif
    const u64 t = carry * 19 + l[0];                      // one v_mad_u64_u32
    C25519_COUNT_MAD(1);
    l[0] = (u32)t & M26;
    l[1] += (u32)(t >> 26);
#pragma unroll
    for (int i = 0; i < 10; i++) r.v[i] = l[i];
}

// RUNS: the operand doublings as a low-priority run and a priority dip behind every column (C25519_VOP2_RUN_*): worth
// 1.7 % on the X25519 ladder at four waves per SIMD, costs 1.7 % in the verification walk at two (profiles/r04_ab_prio.txt)
#ifndef C25519_DIP_EVERY
#define C25519_DIP_EVERY 1                 // A/B knob: a dip behind every n-th column of a product with RUNS
#endif
template <bool RUNS = false>
C25519_DEV void fe_mul_chained(fe& r, const fe& a, const fe& b)
{
    u32 b19[10], a2[10], l[10];
#pragma unroll
    for (int j = 1; j < 10; j++) b19[j] = b.v[j] * 19u;
    if (RUNS) C25519_MASK_RUN_BEGIN();
#pragma unroll
    for (int i = 1; i < 10; i += 2) a2[i] = dbl32(a.v[i]);
    if (RUNS) C25519_MASK_RUN_END();
    u64 acc = 0;
#pragma unroll
    for (int k = 0; k < 10; k++) {
        u32 x[10], y[10];
#pragma unroll
        for (int i = 0; i < 10; i++) {
            const int j = (k - i + 10) % 10;
            const bool wrap = i > k;
            const bool odd2 = (i & 1) && (j & 1);
            x[i] = odd2 ? a2[i] : a.v[i];
            y[i] = wrap ? b19[j] : b.v[j];
        }
        acc = k == 0 ? mad_chain10_from_zero(x, y) : mad_chain10(acc, x, y);
        if (RUNS && (k % C25519_DIP_EVERY) == C25519_DIP_EVERY - 1) { C25519_MASK_RUN_BEGIN(); C25519_MASK_RUN_END(); }
        l[k] = (u32)acc & fe_mask(k);
        acc >>= fe_w(k);
    }
    fe_finish_chain(r, l, acc);
}

// PLAIN: extra(k) is zero for every k (a bare square), so column 0 starts from nothing
template <bool SCALE2, bool PLAIN = false, bool RUNS = false, typename Extra>
C25519_DEV void fe_sqr_chained(fe& r, const fe& a, Extra extra)
{
    u32 f2[10], f19[10], f38[10], l[10];
    if (RUNS) C25519_MASK_RUN_BEGIN();
#pragma unroll
    for (int i = 0; i < 10; i++) f2[i] = dbl32(a.v[i]);
    if (RUNS) C25519_MASK_RUN_END();
#pragma unroll
    for (int j = 6; j < 10; j += 2) f19[j] = a.v[j] * 19u;
#pragma unroll
    for (int j = 5; j < 10; j += 2) f38[j] = a.v[j] * 38u;
    u64 carry = 0;
#pragma unroll
    for (int k = 0; k < 10; k++) {
        u64 acc = SCALE2 ? 0 : carry + extra(k);
        u32 x[6], y[6];
        int cnt = 0;
#pragma unroll
        for (int i = 0; i < 10; i++) {
            const int j = (k - i + 10) % 10;
            if (i > j) continue;
            const bool wrap = (i + j) >= 10;
            const bool odd2 = (i & 1) && (j & 1);
            // coefficient = (i<j ? 2 : 1) * (odd2 ? 2 : 1) * (wrap ? 19 : 1), split over the two operands so
            // that only 38*f_j (odd j) and 19*f_j (even j) are ever needed
            if (wrap && (j & 1)) {
                x[cnt] = (i < j && (i & 1)) ? f2[i] : a.v[i];
                y[cnt] = f38[j];
            } else {
                x[cnt] = (i < j) ? f2[i] : a.v[i];
                y[cnt] = wrap ? f19[j] : (odd2 ? f2[j] : a.v[j]);
            }
            cnt++;
        }
        const bool from_zero = SCALE2 || (PLAIN && k == 0);
        if (k & 1) {                                      // odd columns have 5 unordered pairs, even ones 6
            const u32 x5[5] = { x[0], x[1], x[2], x[3], x[4] }, y5[5] = { y[0], y[1], y[2], y[3], y[4] };
            acc = from_zero ? mad_chain5_from_zero(x5, y5) : mad_chain5(acc, x5, y5);
        } else {
            acc = from_zero ? mad_chain6_from_zero(x, y) : mad_chain6(acc, x, y);
        }
        if (SCALE2) acc = 2 * acc + carry + extra(k);
        if (RUNS && (k % C25519_DIP_EVERY) == C25519_DIP_EVERY - 1) { C25519_MASK_RUN_BEGIN(); C25519_MASK_RUN_END(); }
        l[k] = (u32)acc & fe_mask(k);
        carry = acc >> fe_w(k);
    }
    fe_finish_chain(r, l, carry);
}

#ifndef C25519_ALL_PRODUCT_RUNS
#define C25519_ALL_PRODUCT_RUNS 0        // A/B knob: the in-product runs in EVERY kernel's products, not the ladder's only
#endif
// r = a * b.   beta_a <= 5, beta_b <= 3.3; r may alias a or b.   (ecp_MulReduce)
C25519_DEV void fe_mul(fe& r, const fe& a, const fe& b)
{
    fe_mul_chained<C25519_ALL_PRODUCT_RUNS != 0>(r, a, b);
}
// ... with the low-priority runs inside (the ladder's products)
C25519_DEV void fe_mul_runs(fe& r, const fe& a, const fe& b)
{
    fe_mul_chained<true>(r, a, b);
}
C25519_DEV void fe_sqr_runs(fe& r, const fe& a)
{
    fe_sqr_chained<false, true, true>(r, a, [](int) -> u64 { return 0; });
}

// r = a^2.   beta_a <= 3.3; r may alias a.   (ecp_SqrReduce)
C25519_DEV void fe_sqr(fe& r, const fe& a)
{
    fe_sqr_chained<false, true, C25519_ALL_PRODUCT_RUNS != 0>(r, a, [](int) -> u64 { return 0; });
}

// r = a^2 - m with the subtraction folded into the carry chain (result reduced).
// beta_a <= 3.3, beta_m <= 2 (bias 4p).
C25519_DEV void fe_sqr_sub(fe& r, const fe& a, const fe& m)
{
    fe_sqr_chained<false, false, C25519_ALL_PRODUCT_RUNS != 0>(r, a, [&](int k) -> u64 { return (u64)(2u * fe_2p(k) - m.v[k]); });
}

// r = 2*a^2 + p - m, folded into the carry chain (result reduced).
// beta_a <= 2.3 (columns are doubled), p any beta < 8, m reduced (bias 2p).
C25519_DEV void fe_sqr2_add_sub(fe& r, const fe& a, const fe& p, const fe& m)
{
    fe_sqr_chained<true, false, C25519_ALL_PRODUCT_RUNS != 0>(r, a, [&](int k) -> u64 { return (u64)(p.v[k] + fe_2p(k) - m.v[k]); });
}

C25519_DEV void fe_sqr_n(fe& r, const fe& a, int n)
{
    fe_sqr(r, a);
    for (int i = 1; i < n; i++) fe_sqr(r, r);
}

// r = a + 121665 * b   (ecp_WordMulAddReduce with a24, curve25519_dh.c:53,:81); any beta, reduced out
C25519_DEV void fe_mul121665_add(fe& r, const fe& a, const fe& b)
{
    // one MAD per limb with the previous limb's carry riding in its addend: b*121665 + a < 2^46 for beta < 8, so a
    // carry is < 2^21 and a.v[i] + carry still fits 32 bits -- no 64-bit carry arithmetic (it was a 64-bit shift and a
    // 64-bit add per limb)
    u32 l[10], carry = 0;
    C25519_COUNT_MAD(10);
#pragma unroll
    for (int i = 0; i < 10; i++) {
        const u64 h = (u64)b.v[i] * 121665u + (u64)(a.v[i] + carry);
        l[i] = (u32)h & fe_mask(i);
        carry = (u32)(h >> fe_w(i));
    }
    const u32 t = l[0] + 19u * carry;                     // < 2^27
    l[0] = t & M26;
    l[1] += t >> 26;
#pragma unroll
    for (int i = 0; i < 10; i++) r.v[i] = l[i];
}

// r = c * a for a small constant c < 2^16 (any beta_a <= 8); reduced out
C25519_DEV void fe_mul_small(fe& r, const fe& a, u32 c)
{
    u32 l[10], carry = 0;                                 // a*c < 2^45: carries < 2^20, chained like above
    C25519_COUNT_MAD(10);
#pragma unroll
    for (int i = 0; i < 10; i++) {
        const u64 h = (u64)a.v[i] * c + carry;
        l[i] = (u32)h & fe_mask(i);
        carry = (u32)(h >> fe_w(i));
    }
    const u32 t = l[0] + 19u * carry;
    l[0] = t & M26;
    l[1] += t >> 26;
#pragma unroll
    for (int i = 0; i < 10; i++) r.v[i] = l[i];
}

// one carry pass over 32-bit limbs: brings any beta < 2^6 back to reduced
C25519_DEV void fe_carry32(fe& r, const fe& a)
{
    u32 h[10];
#pragma unroll
    for (int i = 0; i < 10; i++) h[i] = a.v[i];
#pragma unroll
    for (int i = 0; i < 9; i++) {
        h[i + 1] += h[i] >> fe_w(i);
        h[i] &= fe_mask(i);
    }
    u32 c = h[9] >> 25;
    h[9] &= M25;
    h[0] += c * 19u;
    h[1] += h[0] >> 26;
    h[0] &= M26;
#pragma unroll
    for (int i = 0; i < 10; i++) r.v[i] = h[i];
}

// 256-bit little-endian words -> limbs.  All 256 bits are used: bit 255 counts as 2^255 = 19 mod p
// (the reference does not mask it, curve25519_dh.c:104 / SURVEY.md 3.5).
C25519_DEV void fe_from_words(fe& r, const u32 (&w)[8])
{
    r.v[0] = (w[0] & M26) + 19u * (w[7] >> 31);
    r.v[1] = alignbit32(w[1], w[0], 26) & M25;
    r.v[2] = alignbit32(w[2], w[1], 19) & M26;
    r.v[3] = alignbit32(w[3], w[2], 13) & M25;
    r.v[4] = w[3] >> 6;
    r.v[5] = w[4] & M25;
    r.v[6] = alignbit32(w[5], w[4], 25) & M26;
    r.v[7] = alignbit32(w[6], w[5], 19) & M25;
    r.v[8] = alignbit32(w[7], w[6], 12) & M26;
    r.v[9] = (w[7] >> 6) & M25;
}

// same conversion with plain shifts, usable on compile-time constants (folds to immediates)
C25519_DEV constexpr fe fe_const(const u32 (&w)[8])
{
    fe r{};
    const u64 w01 = ((u64)w[1] << 32) | w[0], w12 = ((u64)w[2] << 32) | w[1], w23 = ((u64)w[3] << 32) | w[2];
    const u64 w45 = ((u64)w[5] << 32) | w[4], w56 = ((u64)w[6] << 32) | w[5], w67 = ((u64)w[7] << 32) | w[6];
    r.v[0] = (w[0] & M26) + 19u * (w[7] >> 31);
    r.v[1] = (u32)(w01 >> 26) & M25;
    r.v[2] = (u32)(w12 >> 19) & M26;
    r.v[3] = (u32)(w23 >> 13) & M25;
    r.v[4] = w[3] >> 6;
    r.v[5] = w[4] & M25;
    r.v[6] = (u32)(w45 >> 25) & M26;
    r.v[7] = (u32)(w56 >> 19) & M25;
    r.v[8] = (u32)(w67 >> 12) & M26;
    r.v[9] = (w[7] >> 6) & M25;
    return r;
}

// limbs -> canonical value in [0, p) as 256-bit little-endian words   (ecp_Mod + ecp_WordsToBytes)
C25519_DEV void fe_to_words(u32 (&w)[8], const fe& a)
{
    u32 h[10];
#pragma unroll
    for (int i = 0; i < 10; i++) h[i] = a.v[i];
    // two full carry passes: afterwards every limb is strictly below 2^w, value < 2^255
#pragma unroll
    for (int pass = 0; pass < 2; pass++) {
#pragma unroll
        for (int i = 0; i < 9; i++) {
            h[i + 1] += h[i] >> fe_w(i);
            h[i] &= fe_mask(i);
        }
        u32 c = h[9] >> 25;
        h[9] &= M25;
        h[0] += c * 19u;
    }
    // h0 may now be up to 2^26 + 18: one more ripple of a single possible carry
#pragma unroll
    for (int i = 0; i < 9; i++) {
        h[i + 1] += h[i] >> fe_w(i);
        h[i] &= fe_mask(i);
    }
    // (h9 cannot overflow here: it was < 2^25 and the ripple only happens when the lower limbs were
    //  all-ones, in which case the earlier wrap left h0 small)  -- still fold defensively
    {
        u32 c = h[9] >> 25;
        h[9] &= M25;
        h[0] += c * 19u;
    }
    // q = 1 iff value >= p  <=>  value + 19 >= 2^255
    u32 q = (h[0] + 19u) >> 26;
#pragma unroll
    for (int i = 1; i < 10; i++) q = (h[i] + q) >> fe_w(i);
    h[0] += 19u * q;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        h[i + 1] += h[i] >> fe_w(i);
        h[i] &= fe_mask(i);
    }
    h[9] &= M25;                                           // drops 2^255 * q

    w[0] = h[0] | (h[1] << 26);
    w[1] = (h[1] >> 6) | (h[2] << 19);
    w[2] = (h[2] >> 13) | (h[3] << 13);
    w[3] = (h[3] >> 19) | (h[4] << 6);
    w[4] = h[5] | (h[6] << 25);
    w[5] = (h[6] >> 7) | (h[7] << 19);
    w[6] = (h[7] >> 13) | (h[8] << 12);
    w[7] = (h[8] >> 20) | (h[9] << 6);
}

// x^(2^250 - 1) and x^11: shared front of the two fixed addition chains (254 S + 11 M in total,
// the same operation count as ecp_Inverse :340-409 / ecp_ModExp2523 :116-135)
C25519_DEV void fe_chain250(fe& x250, fe& x11, const fe& x)
{
    fe x2, x9, x5, x10, x20, x50, x100, t;
    fe_sqr(x2, x);
    fe_sqr_n(t, x2, 2);   fe_mul(x9, t, x);
    fe_mul(x11, x9, x2);
    fe_sqr(t, x11);       fe_mul(x5, t, x9);
    fe_sqr_n(t, x5, 5);   fe_mul(x10, t, x5);
    fe_sqr_n(t, x10, 10); fe_mul(x20, t, x10);
    fe_sqr_n(t, x20, 20); fe_mul(t, t, x20);
    fe_sqr_n(t, t, 10);   fe_mul(x50, t, x10);
    fe_sqr_n(t, x50, 50); fe_mul(x100, t, x50);
    fe_sqr_n(t, x100, 100); fe_mul(t, t, x100);
    fe_sqr_n(t, t, 50);   fe_mul(x250, t, x50);
}

// r = z^(p-2); z = 0 gives 0 (what makes low-order X25519 inputs come out as all-zero bytes): the reference's inversion
// (ecp_Inverse, curve25519_mehdi.c:340-409), ~25 000 instructions on one lane
C25519_DEV void fe_invert_fermat(fe& r, const fe& z)
{
    fe x250, x11;
    fe_chain250(x250, x11, z);
    fe_sqr_n(x250, x250, 5);
    fe_mul(r, x250, x11);
}

// the same value by 600 constant-time division steps (safegcd25519.cuh), ~16 000 instructions; 0 gives 0
C25519_DEV void fe_invert_safegcd(fe& r, const fe& z)
{
    u32 w[8], o[8];
    fe_to_words(w, z);
    sg_invert_words(o, w);
    fe_from_words(r, o);
}

// r = 1 / z, 0 for z = 0.  Build knob C25519_INVERT_SAFEGCD = 0: the reference's exponentiation everywhere (A/B).
#ifndef C25519_INVERT_SAFEGCD
#define C25519_INVERT_SAFEGCD 1
#endif
C25519_DEV void fe_invert(fe& r, const fe& z)
{
#if C25519_INVERT_SAFEGCD
    fe_invert_safegcd(r, z);
#else
    fe_invert_fermat(r, z);
#endif
}

// r = 1 / z where the four lanes of an aligned quad (lane & 3; all active) hold the same z: the division steps' three pairs on
// three lanes (sg_divsteps30_quad) -- the operations that have a quad (quad25519.cuh) or a wave (coop25519.cuh) to themselves
C25519_DEV void fe_invert_quad(fe& r, const fe& z)
{
#if C25519_INVERT_SAFEGCD
    u32 w[8], o[8];
    fe_to_words(w, z);
    sg_invert_words_quad(o, w);
    fe_from_words(r, o);
#else
    fe_invert_fermat(r, z);
#endif
}

// r = x^((p-5)/8) = x^(2^252 - 3)
C25519_DEV void fe_pow2523(fe& r, const fe& x)
{
    fe x250, x11;
    fe_chain250(x250, x11, x);
    fe_sqr_n(x250, x250, 2);
    fe_mul(r, x250, x);
}

}  // namespace c25519
