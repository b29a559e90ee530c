/*
 * oracle/orc_sc.c -- arithmetic mod L (the base point order), fold recoding, secret-key clamp.
 * TEST INFRASTRUCTURE ONLY.
 *
 * L = 2^252 + c, c = 0x14DEF9DEA2F79CD65812631A5CF5D3ED.  The reference reduces a (256+w)-bit value
 * Horner-style: fold the top word b with  Y = X - b*(-2^256 mod L)  and add L back on borrow
 * (curve25519_order.c:80-107); -2^256 mod L = 16*c.  Here the word is 64 bits wide.
 */
#include "orc25519.h"
#include <string.h>

typedef unsigned __int128 u128;

static const uint64_t L_[4]  = { 0x5812631A5CF5D3EDull, 0x14DEF9DEA2F79CD6ull, 0, 0x1000000000000000ull };
/* 16*c = -2^256 mod L, 129 bits */
static const uint64_t MR[3]  = { 0x812631A5CF5D3ED0ull, 0x4DEF9DEA2F79CD65ull, 1 };

/* y = [b : x] mod-ish L: y = x - b*MR (+ L on borrow).  curve25519_order.c:80-107 */
static void sc_reduce_hi(uint64_t y[4], uint64_t b, const uint64_t x[4])
{
    uint64_t t[4];
    u128 acc = (u128)b * MR[0];
    t[0] = (uint64_t)acc;
    acc = (u128)b * MR[1] + (uint64_t)(acc >> 64);
    t[1] = (uint64_t)acc;
    acc = (u128)b * MR[2] + (uint64_t)(acc >> 64);
    t[2] = (uint64_t)acc;
    t[3] = (uint64_t)(acc >> 64);

    uint64_t r[4], bw = 0;
    for (int i = 0; i < 4; i++) {
        u128 d = (u128)x[i] - t[i] - bw;
        r[i] = (uint64_t)d;
        bw = (uint64_t)(d >> 64) & 1;
    }
    uint64_t m = (uint64_t)0 - bw;
    u128 c = 0;
    for (int i = 0; i < 4; i++) {
        c = (u128)r[i] + (L_[i] & m) + (uint64_t)(c >> 64);
        y[i] = (uint64_t)c;
    }
}

void orc_sc_mod(uint64_t x[4])                      /* eco_Mod curve25519_order.c:125-129 */
{
    uint64_t n = x[3] >> 60;                        /* x / 2^252, 0..15 */
    uint64_t t[4], r[4], bw = 0;
    u128 acc = 0;
    for (int i = 0; i < 4; i++) {                   /* t = n*L (the reference's _w_NxBPO[n]) */
        acc = (u128)L_[i] * n + (uint64_t)(acc >> 64);
        t[i] = (uint64_t)acc;
    }
    for (int i = 0; i < 4; i++) {
        u128 d = (u128)x[i] - t[i] - bw;
        r[i] = (uint64_t)d;
        bw = (uint64_t)(d >> 64) & 1;
    }
    uint64_t m = (uint64_t)0 - bw;
    u128 c = 0;
    for (int i = 0; i < 4; i++) {
        c = (u128)r[i] + (L_[i] & m) + (uint64_t)(c >> 64);
        x[i] = (uint64_t)c;
    }
}

static void sc_reduce512(uint64_t y[4], uint64_t t[8])
{
    sc_reduce_hi(t + 3, t[7], t + 3);
    sc_reduce_hi(t + 2, t[6], t + 2);
    sc_reduce_hi(t + 1, t[5], t + 1);
    sc_reduce_hi(y, t[4], t);
}

void orc_sc_from_digest(uint64_t y[4], const uint8_t md[64])   /* eco_DigestToWords :139-155 */
{
    uint64_t t[8];
    for (int i = 0; i < 8; i++) {
        uint64_t w = 0;
        for (int j = 7; j >= 0; j--) w = (w << 8) | md[8 * i + j];
        t[i] = w;
    }
    sc_reduce512(y, t);
}

void orc_sc_mul(uint64_t z[4], const uint64_t x[4], const uint64_t y[4])  /* eco_MulReduce :110-122 */
{
    uint64_t t[8] = {0};
    for (int i = 0; i < 4; i++) {
        uint64_t c = 0;
        for (int j = 0; j < 4; j++) {
            u128 p = (u128)x[i] * y[j] + t[i + j] + c;
            t[i + j] = (uint64_t)p;
            c = (uint64_t)(p >> 64);
        }
        t[i + 4] = c;
    }
    sc_reduce512(z, t);
}

void orc_sc_add(uint64_t z[4], const uint64_t x[4], const uint64_t y[4])  /* eco_AddReduce :132-136 */
{
    uint64_t r[4];
    u128 c = 0;
    for (int i = 0; i < 4; i++) {
        c = (u128)x[i] + y[i] + (uint64_t)(c >> 64);
        r[i] = (uint64_t)c;
    }
    sc_reduce_hi(z, (uint64_t)(c >> 64), r);
}

/* cut[n] bit j = scalar bit 32j+31-n  (curve25519_utils.c:144-153) */
void orc_fold8(uint8_t cut[32], const uint64_t k[4])
{
    for (int n = 0; n < 32; n++) {
        unsigned v = 0;
        for (int j = 7; j >= 0; j--) {
            int bit = 32 * j + 31 - n;
            v = (v << 1) | (unsigned)((k[bit >> 6] >> (bit & 63)) & 1);
        }
        cut[n] = (uint8_t)v;
    }
}

/* cut[n] bit j = scalar bit 64j+63-n, n = 0..63  (curve25519_utils.c:125-142) */
void orc_fold4(uint8_t cut[64], const uint64_t k[4])
{
    for (int n = 0; n < 64; n++) {
        unsigned v = 0;
        for (int j = 3; j >= 0; j--) v = (v << 1) | (unsigned)((k[j] >> (63 - n)) & 1);
        cut[n] = (uint8_t)v;
    }
}

void orc_x25519_clamp(uint8_t sk[32])               /* ecp_TrimSecretKey curve25519_utils.c:28-32 */
{
    sk[0] &= 0xf8;
    sk[31] = (uint8_t)((sk[31] | 0x40) & 0x7f);
}

static uint64_t splitmix64(uint64_t *s)
{
    uint64_t z = (*s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

void orc_fill_random(uint8_t *dst, size_t nbytes, uint64_t seed)
{
    uint64_t s = seed;
    size_t i = 0;
    while (i < nbytes) {
        uint64_t v = splitmix64(&s);
        for (int j = 0; j < 8 && i < nbytes; j++, i++) dst[i] = (uint8_t)(v >> (8 * j));
    }
}
