// curve25519_amd/csrc/sc25519.cuh -- scalars mod L (the base-point order) and fold recoding, per lane.
//
// Device replacement for source/curve25519_order.c (eco_ReduceHiWord :80, eco_MulReduce :110,
// eco_Mod :125, eco_AddReduce :132, eco_DigestToWords :139) and for ecp_8Folds / ecp_4Folds
// (source/curve25519_utils.c:144 / :125).  Scalars are 8 x 32-bit little-endian words.  Only canonical
// results are ever exported, so any correct reduction gives the reference's bytes; the structure here is
// the reference's Horner fold of one top word at a time with -2^256 mod L = 16c (129 bits).
#pragma once
#include "curve_constants.cuh"
#include "fe25519.cuh"

namespace c25519 {

// y[0..7] = ([b : x[0..7]]) reduced to 256 bits (congruent mod L, not canonical):
//   y = x - b * (16c), plus L when that borrows.   x and y may alias.
C25519_DEV void sc_reduce_hi(u32* y, u32 b, const u32* x)
{
    u32 t[6];
    u64 acc = 0;
#pragma unroll
    for (int i = 0; i < 5; i++) {
        acc += (u64)b * K_MINUS_R[i];
        t[i] = (u32)acc;
        acc >>= 32;
    }
    t[5] = (u32)acc;

    u32 r[8];
    u32 borrow = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const u64 d = (u64)x[i] - (i < 6 ? t[i] : 0u) - borrow;
        r[i] = (u32)d;
        borrow = (u32)(d >> 63);
    }
    const u32 m = 0u - borrow;
    u64 c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        c += (u64)r[i] + (K_L[i] & m);
        y[i] = (u32)c;
        c >>= 32;
    }
}

// x = x mod L, canonical.  x < 2^256.   (eco_Mod: subtract (x >> 252) * L, add L back on borrow)
C25519_DEV void sc_mod(u32 (&x)[8])
{
    const u32 n = x[7] >> 28;
    u32 r[8];
    u64 mul = 0;
    u32 borrow = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        mul += (u64)K_L[i] * n;
        const u64 d = (u64)x[i] - (u32)mul - borrow;
        mul >>= 32;
        r[i] = (u32)d;
        borrow = (u32)(d >> 63);
    }
    const u32 m = 0u - borrow;
    u64 c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        c += (u64)r[i] + (K_L[i] & m);
        x[i] = (u32)c;
        c >>= 32;
    }
}

// hi[0..N) * 16c -> p[0..N+5)   (16c = -2^256 mod L, 129 bits: K_MINUS_R)
template <int N>
C25519_DEV void sc_times_16c(u32 (&p)[N + 5], const u32* hi)
{
#pragma unroll
    for (int i = 0; i < N + 5; i++) p[i] = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
        u32 carry = 0;
#pragma unroll
        for (int j = 0; j < 5; j++) {
            const u64 q = (u64)hi[i] * K_MINUS_R[j] + p[i + j] + carry;
            p[i + j] = (u32)q;
            carry = (u32)(q >> 32);
        }
        p[i + 5] = carry;
    }
}

// y = t[0..15] mod-ish L (256 bits, not canonical).   (eco_DigestToWords / tail of eco_MulReduce)
// Two folds of everything above bit 256 with 2^256 = -16c (mod L), each kept non-negative by a multiple of L that exceeds what it
// subtracts (L << 134 > 2^385, L << 9 > 2^260; "- p" as "+ ~p + 1" in a fixed number of words), then the split at bit 252 of
// sc_mod: 40 + 25 + 4 multiply-adds and three carry chains, ~250 instructions; the reference's Horner fold of one top word at a
// time (eco_ReduceHiWord eight times: sc_reduce_hi, still what sc_add uses) is ~650.  Any correct reduction gives the reference's
// bytes: only canonical results are exported.
C25519_DEV void sc_reduce512(u32 (&y)[8], u32 (&t)[16])
{
#if defined(C25519_SC_REDUCE_HORNER) && C25519_SC_REDUCE_HORNER      // A/B knob: the reference's fold, one top word at a time
#pragma unroll
    for (int k = 7; k >= 1; k--) sc_reduce_hi(&t[k], t[k + 8], &t[k]);
    sc_reduce_hi(y, t[8], &t[0]);
    return;
#endif
    u32 p1[13], y1[13];
    sc_times_16c<8>(p1, &t[8]);
    u64 c = 1;
#pragma unroll
    for (int k = 0; k < 13; k++) {
        c += (u64)(k < 8 ? t[k] : 0u) + K_L_SHL134[k] + (u32)~p1[k];
        y1[k] = (u32)c;
        c >>= 32;
    }
    u32 p2[10], y2[9];                                     // y1 < 2^387: its part above bit 256 is 131 bits, p2 < 2^260
    sc_times_16c<5>(p2, &y1[8]);
    c = 1;
#pragma unroll
    for (int k = 0; k < 9; k++) {
        c += (u64)(k < 8 ? y1[k] : 0u) + K_L_SHL9[k] + (u32)~p2[k];
        y2[k] = (u32)c;
        c >>= 32;
    }
    // y2 < 2^262:  (y2 mod 2^252) - (y2 >> 252) * c, plus L when that borrows   (c = L - 2^252: the low four words of K_L)
    const u32 n = (y2[7] >> 28) | (y2[8] << 4);
    y2[7] &= 0x0fffffffu;
    u32 r[8];
    u64 mul = 0;
    u32 borrow = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        mul += (u64)(i < 4 ? K_L[i] : 0u) * n;
        const u64 d = (u64)y2[i] - (u32)mul - borrow;
        mul >>= 32;
        r[i] = (u32)d;
        borrow = (u32)(d >> 63);
    }
    const u32 m = 0u - borrow;
    c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        c += (u64)r[i] + (K_L[i] & m);
        y[i] = (u32)c;
        c >>= 32;
    }
}

// z = x * y mod-ish L
C25519_DEV void sc_mul(u32 (&z)[8], const u32 (&x)[8], const u32 (&y)[8])
{
    u32 t[16];
#pragma unroll
    for (int i = 0; i < 16; i++) t[i] = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        u32 carry = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const u64 p = (u64)x[i] * y[j] + t[i + j] + carry;
            t[i + j] = (u32)p;
            carry = (u32)(p >> 32);
        }
        t[i + 8] = carry;
    }
    sc_reduce512(z, t);
}

// z = x + y mod-ish L
C25519_DEV void sc_add(u32 (&z)[8], const u32 (&x)[8], const u32 (&y)[8])
{
    u32 r[8];
    u64 c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        c += (u64)x[i] + y[i];
        r[i] = (u32)c;
        c >>= 32;
    }
    sc_reduce_hi(z, (u32)c, r);
}

// 8-fold column for walk step n (n = 0 first): bit j = scalar bit 32j + 31 - n.
// Call with n = 0, 1, ..., 31 in order; k is consumed (shifted left one bit per call).
C25519_DEV u32 fold8_next(u32 (&k)[8])
{
    u32 idx = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        idx |= (k[j] >> 31) << j;
        k[j] <<= 1;
    }
    return idx;
}

// 4-fold column for walk step n (0..63): bit i = bit 63-n of 64-bit limb i, i.e. the odd words for
// n < 32 and the even words afterwards.  Consumes k like fold8_next.
C25519_DEV u32 fold4_next(u32 (&k)[8], bool low_half)
{
    u32 idx = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int w = low_half ? 2 * i : 2 * i + 1;
        idx |= (k[w] >> 31) << i;
        k[w] <<= 1;
    }
    return idx;
}

C25519_DEV void clamp_words(u32 (&k)[8])          // ecp_TrimSecretKey, curve25519_utils.c:28-32
{
    k[0] &= 0xfffffff8u;
    k[7] = (k[7] | 0x40000000u) & 0x7fffffffu;
}

}  // namespace c25519
