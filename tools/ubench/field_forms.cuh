// tools/ubench/field_forms.cuh -- the two field representations the product did NOT choose, written out so that the
// choice is a measurement and not an estimate ("settle the field representation with a real A/B"):
//
//   fe8  saturated radix 2^32, 8 words, value < 2^256 congruent mod p: north_star's shape, the reference's own
//        (ecp_MulReduce curve25519_mehdi.c:278-294 / asm64 Mult.s:98-138, Square.s:31-100): a 32x32 product per
//        v_mad_u64_u32 whose carry-out goes to a third accumulator word with v_addc_co_u32, x38 fold of the high half.
//   fe9  9 unsaturated limbs of radix 2^(255/9) (29/28/28 bits): 81 products per multiplication instead of 100, but
//        19*limb no longer fits 32 bits, so the 2^255 = 19 wrap needs 17 separate columns and a second fold, and the
//        3.67 spare bits per limb do not cover a biased subtraction followed by a product: operands are carried first.
//
// Both are complete enough to run the X25519 ladder (mul, sqr, add, sub, a24 step, select, bytes in/out), are checked
// against the RFC 7748 vectors and the product's field on the CPU (tests/test_field_forms.py) and timed on the GPU by
// tools/ubench/field_ab.hip against the product's radix-2^25.5 form (curve25519_amd/csrc/fe25519.cuh).
// Plain C++ over two primitives: mad96() (v_mad_u64_u32 + v_addc_co_u32) and the product's own.  TOOLING, not product.
#pragma once
#include "fe25519.cuh"

namespace c25519 {

// ===================================================================================================================
// fe8: saturated radix 2^32
// ===================================================================================================================
struct fe8 { u32 v[8]; };

#ifdef C25519_VALU_PRIMITIVES          // host model (tests)
inline void mad96(u64& lo, u32& hi, u32 x, u32 y)
{
    const u64 p = (u64)x * y, r = lo + p;
    hi += r < lo;
    lo = r;
}
#else
C25519_DEV void mad96(u64& lo, u32& hi, u32 x, u32 y)
{
    asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(lo), "+v"(hi) : "v"(x), "v"(y) : "vcc");
}
#endif

C25519_DEV void fe8_from_words(fe8& r, const u32 (&w)[8])
{
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = w[i];
}

// t[0..7] + carry*2^256 -> 8 words congruent mod p: fold carry*38 twice (second fold cannot carry out again)
C25519_DEV void fe8_fold(fe8& r, u32 (&t)[8], u64 carry)
{
    u64 d = (u64)t[0] + carry * 38;
    r.v[0] = (u32)d; d >>= 32;
#pragma unroll
    for (int i = 1; i < 8; i++) { d += t[i]; r.v[i] = (u32)d; d >>= 32; }
    r.v[0] += 38u * (u32)d;
}

C25519_DEV void fe8_add(fe8& r, const fe8& a, const fe8& b)
{
    u32 t[8];
    u64 c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) { c += (u64)a.v[i] + b.v[i]; t[i] = (u32)c; c >>= 32; }
    fe8_fold(r, t, c);
}

C25519_DEV void fe8_sub(fe8& r, const fe8& a, const fe8& b)
{
    u32 t[8];
    u64 bw = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) { const u64 d = (u64)a.v[i] - b.v[i] - bw; t[i] = (u32)d; bw = (d >> 63) & 1u; }
    // borrow: subtract 38 (2^256 = 38), twice at most
    u64 d = (u64)t[0] - 38 * bw;
    r.v[0] = (u32)d; u64 b2 = (d >> 63) & 1u;
#pragma unroll
    for (int i = 1; i < 8; i++) { d = (u64)t[i] - b2; r.v[i] = (u32)d; b2 = (d >> 63) & 1u; }
    r.v[0] -= 38u * (u32)b2;
}

// 16 words -> 8 congruent mod p: low + 38 * high
C25519_DEV void fe8_reduce512(fe8& r, const u32 (&t)[16])
{
    u32 l[8];
    u64 c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) { c += (u64)t[8 + i] * 38u + t[i]; l[i] = (u32)c; c >>= 32; }
    fe8_fold(r, l, c);
}

C25519_DEV void fe8_mul(fe8& r, const fe8& a, const fe8& b)
{
    u32 t[16];
    u64 acc = 0;
    u32 hi = 0;
#pragma unroll
    for (int k = 0; k < 15; k++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int j = k - i;
            if (j < 0 || j > 7) continue;
            mad96(acc, hi, a.v[i], b.v[j]);
        }
        t[k] = (u32)acc;
        acc = (acc >> 32) | ((u64)hi << 32);
        hi = 0;
    }
    t[15] = (u32)acc;
    fe8_reduce512(r, t);
}

// 28 cross products once, the 512-bit sum doubled by a one-bit shift, then the 8 squares (the shape of Square.s)
C25519_DEV void fe8_sqr(fe8& r, const fe8& a)
{
    u32 t[16];
    u64 acc = 0;
    u32 hi = 0;
    t[0] = 0;
#pragma unroll
    for (int k = 1; k < 14; k++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int j = k - i;
            if (j <= i || j > 7) continue;
            mad96(acc, hi, a.v[i], a.v[j]);
        }
        t[k] = (u32)acc;
        acc = (acc >> 32) | ((u64)hi << 32);
        hi = 0;
    }
    t[14] = (u32)acc;
    t[15] = (u32)(acc >> 32);
#pragma unroll
    for (int k = 15; k > 0; k--) t[k] = (t[k] << 1) | (t[k - 1] >> 31);
    t[0] = 0;
    u64 c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const u64 sq = (u64)a.v[i] * a.v[i];
        c += (u64)t[2 * i] + (u32)sq;
        t[2 * i] = (u32)c; c >>= 32;
        c += (u64)t[2 * i + 1] + (u32)(sq >> 32);
        t[2 * i + 1] = (u32)c; c >>= 32;
    }
    fe8_reduce512(r, t);
}

// r = a + 121665 * b
C25519_DEV void fe8_mul121665_add(fe8& r, const fe8& a, const fe8& b)
{
    u32 l[8];
    u64 c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) { c += (u64)b.v[i] * 121665u + a.v[i]; l[i] = (u32)c; c >>= 32; }
    fe8_fold(r, l, c);
}

C25519_DEV void fe8_select(fe8& r, u32 mask, const fe8& a, const fe8& b)
{
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = (a.v[i] & mask) | (b.v[i] & ~mask);
}

C25519_DEV void fe8_sqr_n(fe8& r, const fe8& a, int n)
{
    fe8_sqr(r, a);
    for (int i = 1; i < n; i++) fe8_sqr(r, r);
}

C25519_DEV void fe8_invert(fe8& r, const fe8& z)
{
    fe8 x2, x9, x11, x5, x10, x20, x50, x100, t;
    fe8_sqr(x2, z);
    fe8_sqr_n(t, x2, 2);   fe8_mul(x9, t, z);
    fe8_mul(x11, x9, x2);
    fe8_sqr(t, x11);       fe8_mul(x5, t, x9);
    fe8_sqr_n(t, x5, 5);   fe8_mul(x10, t, x5);
    fe8_sqr_n(t, x10, 10); fe8_mul(x20, t, x10);
    fe8_sqr_n(t, x20, 20); fe8_mul(t, t, x20);
    fe8_sqr_n(t, t, 10);   fe8_mul(x50, t, x10);
    fe8_sqr_n(t, x50, 50); fe8_mul(x100, t, x50);
    fe8_sqr_n(t, x100, 100); fe8_mul(t, t, x100);
    fe8_sqr_n(t, t, 50);   fe8_mul(t, t, x50);
    fe8_sqr_n(t, t, 5);
    fe8_mul(r, t, x11);
}

// canonical words: two conditional subtractions of p (ecp_Mod)
C25519_DEV void fe8_to_words(u32 (&w)[8], const fe8& a)
{
    u32 t[8];
#pragma unroll
    for (int i = 0; i < 8; i++) t[i] = a.v[i];
    // fold bit 255: value = low255 + 19 * bit255
    u64 c = (u64)t[0] + 19u * (t[7] >> 31);
    t[7] &= 0x7fffffffu;
    t[0] = (u32)c; c >>= 32;
#pragma unroll
    for (int i = 1; i < 8; i++) { c += t[i]; t[i] = (u32)c; c >>= 32; }
    // now < 2^255 + 19: once more for the possible new bit 255, then subtract p if >= p
#pragma unroll
    for (int pass = 0; pass < 2; pass++) {
        u64 d = (u64)t[0] + 19u;
        u32 s[8];
        s[0] = (u32)d; d >>= 32;
#pragma unroll
        for (int i = 1; i < 8; i++) { d += t[i]; s[i] = (u32)d; d >>= 32; }
        const u32 ge = 0u - (s[7] >> 31);                 // t + 19 >= 2^255  <=>  t >= p
        s[7] &= 0x7fffffffu;
#pragma unroll
        for (int i = 0; i < 8; i++) t[i] = (s[i] & ge) | (t[i] & ~ge);
    }
#pragma unroll
    for (int i = 0; i < 8; i++) w[i] = t[i];
}

// ===================================================================================================================
// fe9: nine limbs of radix 2^(255/9)
// ===================================================================================================================
struct fe9 { u32 v[9]; };
C25519_DEV constexpr int fe9_w(int i) { return (i % 3 == 0) ? 29 : 28; }
C25519_DEV constexpr int fe9_pos(int i) { return (85 * i + 2) / 3; }                  // ceil(255 i / 9)
C25519_DEV constexpr u32 fe9_mask(int i) { return (1u << fe9_w(i)) - 1u; }
// a_i * b_j lands one bit above limb (i+j)'s position when (i%3, j%3) is (1,1), (1,2) or (2,1)
C25519_DEV constexpr bool fe9_dbl(int i, int j) { return (i % 3 == 1 && j % 3 != 0) || (i % 3 == 2 && j % 3 == 1); }
// limbs of 2p: (2^30-38, 2^29-2, 2^29-2, 2^30-2, ...)
C25519_DEV constexpr u32 fe9_2p(int i) { return i == 0 ? 0x3fffffdau : (2u << fe9_w(i)) - 2u; }

C25519_DEV void fe9_from_words(fe9& r, const u32 (&w)[8])
{
    // all 256 bits, bit 255 counted as 19
#pragma unroll
    for (int i = 0; i < 9; i++) {
        const int p = fe9_pos(i), word = p / 32, sh = p % 32;
        u64 x = w[word];
        if (word + 1 < 8) x |= (u64)w[word + 1] << 32;
        r.v[i] = (u32)(x >> sh) & fe9_mask(i);
    }
    r.v[0] += 19u * (w[7] >> 31);
}

// one carry pass: any limb < 2^32 -> limbs < 2^w + small
C25519_DEV void fe9_carry(fe9& r, const fe9& a)
{
    u32 h[9];
#pragma unroll
    for (int i = 0; i < 9; i++) h[i] = a.v[i];
#pragma unroll
    for (int i = 0; i < 8; i++) { h[i + 1] += h[i] >> fe9_w(i); h[i] &= fe9_mask(i); }
    const u32 c = h[8] >> 28;
    h[8] &= fe9_mask(8);
    h[0] += 19u * c;
    h[1] += h[0] >> 29;
    h[0] &= fe9_mask(0);
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = h[i];
}

C25519_DEV void fe9_add(fe9& r, const fe9& a, const fe9& b)
{
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = a.v[i] + b.v[i];
}
C25519_DEV void fe9_sub(fe9& r, const fe9& a, const fe9& b)            // b reduced
{
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = a.v[i] + fe9_2p(i) - b.v[i];
}

// 17 column sums -> reduced limbs: carry the high half into limbs h_k (so that 19*h_k fits), c_k = S_k + 19 h_k, carry
C25519_DEV void fe9_fold_columns(fe9& r, u64 (&S)[17])
{
    u32 h[8];
    u64 c = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {                         // columns 9..16 sit at limb positions 0..7 (+255)
        c += S[9 + k];
        h[k] = (u32)c & fe9_mask(k);
        c >>= fe9_w(k);
    }
    // c now has the weight of limb 8 (+255): goes into column 8 times 19
    u64 acc = 0;
    u32 l[9];
#pragma unroll
    for (int k = 0; k < 9; k++) {
        acc += S[k] + (k < 8 ? (u64)h[k] * 19u : c * 19u);
        l[k] = (u32)acc & fe9_mask(k);
        acc >>= fe9_w(k);
    }
    const u64 t = acc * 19u + l[0];
    l[0] = (u32)t & fe9_mask(0);
    l[1] += (u32)(t >> 29);
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = l[i];
}

// beta_a * beta_b <= 3.5 (column sums below 2^64): callers carry biased differences first
C25519_DEV void fe9_mul(fe9& r, const fe9& a, const fe9& b)
{
    u32 a2[9];
#pragma unroll
    for (int i = 0; i < 9; i++) a2[i] = dbl32(a.v[i]);
    u64 S[17];
#pragma unroll
    for (int k = 0; k < 17; k++) {
        u64 acc = 0;
#pragma unroll
        for (int i = 0; i < 9; i++) {
            const int j = k - i;
            if (j < 0 || j > 8) continue;
            acc += (u64)(fe9_dbl(i, j) ? a2[i] : a.v[i]) * b.v[j];
        }
        S[k] = acc;
    }
    fe9_fold_columns(r, S);
}

C25519_DEV void fe9_sqr(fe9& r, const fe9& a)
{
    u32 a2[9], a4[9];
#pragma unroll
    for (int i = 0; i < 9; i++) { a2[i] = dbl32(a.v[i]); a4[i] = dbl32(a2[i]); }
    u64 S[17];
#pragma unroll
    for (int k = 0; k < 17; k++) {
        u64 acc = 0;
#pragma unroll
        for (int i = 0; i < 9; i++) {
            const int j = k - i;
            if (j < i || j > 8) continue;
            const bool d = fe9_dbl(i, j);
            const u32 x = (i < j) ? (d ? a4[i] : a2[i]) : (d ? a2[i] : a.v[i]);
            acc += (u64)x * a.v[j];
        }
        S[k] = acc;
    }
    fe9_fold_columns(r, S);
}

C25519_DEV void fe9_mul121665_add(fe9& r, const fe9& a, const fe9& b)
{
    u64 acc = 0;
    u32 l[9];
#pragma unroll
    for (int k = 0; k < 9; k++) {
        acc += (u64)b.v[k] * 121665u + a.v[k];
        l[k] = (u32)acc & fe9_mask(k);
        acc >>= fe9_w(k);
    }
    const u64 t = acc * 19u + l[0];
    l[0] = (u32)t & fe9_mask(0);
    l[1] += (u32)(t >> 29);
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = l[i];
}

C25519_DEV void fe9_select(fe9& r, u32 mask, const fe9& a, const fe9& b)
{
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = (a.v[i] & mask) | (b.v[i] & ~mask);
}

C25519_DEV void fe9_sqr_n(fe9& r, const fe9& a, int n)
{
    fe9_sqr(r, a);
    for (int i = 1; i < n; i++) fe9_sqr(r, r);
}

C25519_DEV void fe9_invert(fe9& r, const fe9& z)
{
    fe9 x2, x9, x11, x5, x10, x20, x50, x100, t;
    fe9_sqr(x2, z);
    fe9_sqr_n(t, x2, 2);   fe9_mul(x9, t, z);
    fe9_mul(x11, x9, x2);
    fe9_sqr(t, x11);       fe9_mul(x5, t, x9);
    fe9_sqr_n(t, x5, 5);   fe9_mul(x10, t, x5);
    fe9_sqr_n(t, x10, 10); fe9_mul(x20, t, x10);
    fe9_sqr_n(t, x20, 20); fe9_mul(t, t, x20);
    fe9_sqr_n(t, t, 10);   fe9_mul(x50, t, x10);
    fe9_sqr_n(t, x50, 50); fe9_mul(x100, t, x50);
    fe9_sqr_n(t, x100, 100); fe9_mul(t, t, x100);
    fe9_sqr_n(t, t, 50);   fe9_mul(t, t, x50);
    fe9_sqr_n(t, t, 5);
    fe9_mul(r, t, x11);
}

C25519_DEV void fe9_to_words(u32 (&w)[8], const fe9& a)
{
    fe9 t;
    fe9_carry(t, a);
    fe9_carry(t, t);
    u32 h[9];
#pragma unroll
    for (int i = 0; i < 9; i++) h[i] = t.v[i];
    // full ripple so that every limb is strictly below 2^w, then q = (value >= p)
#pragma unroll
    for (int pass = 0; pass < 2; pass++) {
#pragma unroll
        for (int i = 0; i < 8; i++) { h[i + 1] += h[i] >> fe9_w(i); h[i] &= fe9_mask(i); }
        const u32 c = h[8] >> 28;
        h[8] &= fe9_mask(8);
        h[0] += 19u * c;
    }
    u32 q = (h[0] + 19u) >> 29;
#pragma unroll
    for (int i = 1; i < 9; i++) q = (h[i] + q) >> fe9_w(i);
    h[0] += 19u * q;
#pragma unroll
    for (int i = 0; i < 8; i++) { h[i + 1] += h[i] >> fe9_w(i); h[i] &= fe9_mask(i); }
    h[8] &= fe9_mask(8);
    // pack
    u64 acc = 0;
    int bits = 0, word = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        acc |= (u64)h[i] << bits;
        bits += fe9_w(i);
        if (bits >= 32) { w[word++] = (u32)acc; acc >>= 32; bits -= 32; }
    }
    if (word < 8) w[word] = (u32)acc;
}

// ===================================================================================================================
// the same X25519 ladder step (curve25519_dh.c:57-84: 5 M + 4 S + a24 + 8 add/sub) in each form
// ===================================================================================================================
// product's form: ladder_step() of x25519.cuh.

C25519_DEV void ladder_step8(fe8& SX, fe8& SZ, fe8& DX, fe8& DZ, const fe8& base, u32 prev_eq)
{
    fe8 A, B, C, Dp, P, M;
    fe8_sub(A, SX, SZ);
    fe8_add(B, SX, SZ);
    fe8_sub(C, DX, DZ);
    fe8_add(Dp, DX, DZ);
    fe8_select(P, prev_eq, Dp, B);
    fe8_select(M, prev_eq, C, A);
    fe8_mul(A, A, Dp);
    fe8_mul(B, C, B);
    fe8_add(C, A, B);
    fe8_sub(B, A, B);
    fe8_sqr(SX, C);
    fe8_sqr(A, B);
    fe8_mul(SZ, A, base);
    fe8_sqr(A, P);
    fe8_sqr(B, M);
    fe8_mul(DX, A, B);
    fe8_sub(B, A, B);
    fe8_mul121665_add(A, A, B);
    fe8_mul(DZ, B, A);
}

C25519_DEV void ladder_step9(fe9& SX, fe9& SZ, fe9& DX, fe9& DZ, const fe9& base, u32 prev_eq)
{
    fe9 A, B, C, Dp, P, M, t;
    fe9_sub(t, SX, SZ);  fe9_carry(A, t);          // beta 3 -> 1: a product with a beta-2 sum must stay below 3.5
    fe9_add(B, SX, SZ);
    fe9_sub(t, DX, DZ);  fe9_carry(C, t);
    fe9_add(Dp, DX, DZ);
    fe9_select(t, prev_eq, Dp, B);  fe9_carry(P, t);   // squaring needs beta^2 <= 3.5
    fe9_select(M, prev_eq, C, A);
    fe9_mul(A, A, Dp);
    fe9_mul(B, C, B);
    fe9_add(t, A, B);    fe9_carry(C, t);
    fe9_sub(t, A, B);    fe9_carry(B, t);
    fe9_sqr(SX, C);
    fe9_sqr(A, B);
    fe9_mul(SZ, A, base);
    fe9_sqr(A, P);
    fe9_sqr(B, M);
    fe9_mul(DX, A, B);
    fe9_sub(t, A, B);    fe9_carry(B, t);
    fe9_mul121665_add(A, A, B);
    fe9_mul(DZ, B, A);
}

// full X25519 in the two alternative forms (validation of the forms: tests/test_field_forms.py)
template <typename F>
struct form_ops;
template <> struct form_ops<fe8> {
    static C25519_DEV void from_words(fe8& r, const u32 (&w)[8]) { fe8_from_words(r, w); }
    static C25519_DEV void one(fe8& r) { const u32 w[8] = { 1, 0, 0, 0, 0, 0, 0, 0 }; fe8_from_words(r, w); }
    static C25519_DEV void step(fe8& a, fe8& b, fe8& c, fe8& d, const fe8& x, u32 eq) { ladder_step8(a, b, c, d, x, eq); }
    static C25519_DEV void select(fe8& r, u32 m, const fe8& a, const fe8& b) { fe8_select(r, m, a, b); }
    static C25519_DEV void invert(fe8& r, const fe8& a) { fe8_invert(r, a); }
    static C25519_DEV void mul(fe8& r, const fe8& a, const fe8& b) { fe8_mul(r, a, b); }
    static C25519_DEV void to_words(u32 (&w)[8], const fe8& a) { fe8_to_words(w, a); }
};
template <> struct form_ops<fe9> {
    static C25519_DEV void from_words(fe9& r, const u32 (&w)[8]) { fe9_from_words(r, w); }
    static C25519_DEV void one(fe9& r) { const u32 w[8] = { 1, 0, 0, 0, 0, 0, 0, 0 }; fe9_from_words(r, w); }
    static C25519_DEV void step(fe9& a, fe9& b, fe9& c, fe9& d, const fe9& x, u32 eq) { ladder_step9(a, b, c, d, x, eq); }
    static C25519_DEV void select(fe9& r, u32 m, const fe9& a, const fe9& b) { fe9_select(r, m, a, b); }
    static C25519_DEV void invert(fe9& r, const fe9& a) { fe9_invert(r, a); }
    static C25519_DEV void mul(fe9& r, const fe9& a, const fe9& b) { fe9_mul(r, a, b); }
    static C25519_DEV void to_words(u32 (&w)[8], const fe9& a) { fe9_to_words(w, a); }
};

// clamp(k) * u through the form's ladder step, 255 steps from (D, S) = (O, u): the textbook ladder in the
// (sum, double) bookkeeping of x25519.cuh -- S' = S + D (difference u), D' = 2 * (the multiple the bit selects), with
// prev_eq telling ladder_step whether that multiple currently sits in D or in S.
template <typename F>
C25519_DEV void x25519_form(u32 (&out)[8], const u32 (&u)[8], const u32 (&kin)[8])
{
    typedef form_ops<F> O;
    u32 k[8];
#pragma unroll
    for (int i = 0; i < 8; i++) k[i] = kin[i];
    k[0] &= 0xfffffff8u;
    k[7] = (k[7] | 0x40000000u) & 0x7fffffffu;
    const u32 zero[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
    F X1, SX, SZ, DX, DZ, PX, PZ, zi;
    O::from_words(X1, u);
    O::one(DX); O::from_words(DZ, zero);          // D = O = (1 : 0)
    SX = X1; O::one(SZ);                          // S = (u : 1)
    u32 prev = 0;
#pragma unroll 1
    for (int b = 254; b >= 0; b--) {
        const u32 bit = (k[b >> 5] >> (b & 31)) & 1u;
        O::step(SX, SZ, DX, DZ, X1, (u32)0 - (u32)(bit == prev));
        prev = bit;
    }
    const u32 m = (u32)0 - prev;
    O::select(PX, m, SX, DX);
    O::select(PZ, m, SZ, DZ);
    O::invert(zi, PZ);
    O::mul(PX, PX, zi);
    O::to_words(out, PX);
}

}  // namespace c25519
