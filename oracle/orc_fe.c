/*
 * oracle/orc_fe.c -- GF(2^255-19) arithmetic, CPU restatement (TEST INFRASTRUCTURE ONLY).
 *
 * Restates the contract of the reference's source/curve25519_mehdi.c: every *Reduce style
 * operation accepts any 256-bit inputs and returns a 256-bit value congruent mod p (possibly >= p);
 * only orc_fe_mod() canonicalises.  2^256 == 38 (mod p) is the folding constant, and two folds
 * always suffice (curve25519_mehdi.c:137-157, :255-273).
 */
#include "orc25519.h"
#include <string.h>

typedef unsigned __int128 u128;

void orc_fe_frombytes(orc_fe y, const uint8_t *x)      /* ecp_BytesToWords curve25519_utils.c:43 */
{
    for (int i = 0; i < 4; i++) {
        uint64_t w = 0;
        for (int j = 7; j >= 0; j--) w = (w << 8) | x[8 * i + j];
        y[i] = w;
    }
}

void orc_fe_tobytes(uint8_t *y, const orc_fe x)        /* ecp_WordsToBytes curve25519_utils.c:61 */
{
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 8; j++) y[8 * i + j] = (uint8_t)(x[i] >> (8 * j));
}

/* add the small value `extra` (< 2^70) * 1 into z, then fold the single possible carry-out */
static inline void fold_in(orc_fe z, u128 extra)
{
    u128 t = (u128)z[0] + (uint64_t)extra;
    z[0] = (uint64_t)t;
    t = (u128)z[1] + (uint64_t)(extra >> 64) + (uint64_t)(t >> 64);
    z[1] = (uint64_t)t;
    t = (u128)z[2] + (uint64_t)(t >> 64);
    z[2] = (uint64_t)t;
    t = (u128)z[3] + (uint64_t)(t >> 64);
    z[3] = (uint64_t)t;
    /* a second wrap leaves a tiny residue, so +38 cannot carry again */
    z[0] += 38 * (uint64_t)(t >> 64);
}

void orc_fe_add(orc_fe z, const orc_fe x, const orc_fe y)   /* curve25519_mehdi.c:134-158 */
{
    u128 t = 0;
    uint64_t r[4];
    for (int i = 0; i < 4; i++) {
        t = (u128)x[i] + y[i] + (uint64_t)(t >> 64);
        r[i] = (uint64_t)t;
    }
    uint64_t c = (uint64_t)(t >> 64);
    memcpy(z, r, sizeof r);
    fold_in(z, (u128)c * 38);
}

void orc_fe_sub(orc_fe z, const orc_fe x, const orc_fe y)   /* curve25519_mehdi.c:161-183 */
{
    uint64_t r[4], b = 0;
    for (int i = 0; i < 4; i++) {
        u128 t = (u128)x[i] - y[i] - b;
        r[i] = (uint64_t)t;
        b = (uint64_t)(t >> 64) & 1;
    }
    /* borrow means we are 2^256 too high: subtract 38, twice at most */
    uint64_t s = 38 * b;
    b = 0;
    for (int i = 0; i < 4; i++) {
        u128 t = (u128)r[i] - s - b;
        r[i] = (uint64_t)t;
        b = (uint64_t)(t >> 64) & 1;
        s = 0;
    }
    r[0] -= 38 * b;
    memcpy(z, r, sizeof r);
}

/* reduce a 512-bit product t[0..7] to 256 bits: lo + 38*hi, two folds */
static inline void reduce512(orc_fe z, const uint64_t t[8])
{
    u128 acc = 0;
    uint64_t r[4];
    for (int i = 0; i < 4; i++) {
        acc = (u128)t[i] + (u128)t[4 + i] * 38 + (uint64_t)(acc >> 64);
        r[i] = (uint64_t)acc;
    }
    uint64_t c = (uint64_t)(acc >> 64);      /* <= 38 */
    memcpy(z, r, sizeof r);
    fold_in(z, (u128)c * 38);
}

void orc_fe_mul(orc_fe z, const orc_fe x, const orc_fe y)   /* curve25519_mehdi.c:278-294 */
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
    reduce512(z, t);
}

void orc_fe_sqr(orc_fe z, const orc_fe x)                   /* curve25519_mehdi.c:310-329 */
{
    orc_fe_mul(z, x, x);
}

void orc_fe_mulw_add(orc_fe z, const orc_fe y, uint64_t b, const orc_fe x) /* curve25519_mehdi.c:243-274 */
{
    u128 acc = 0;
    uint64_t r[4];
    for (int i = 0; i < 4; i++) {
        acc = (u128)x[i] * b + y[i] + (uint64_t)(acc >> 64);
        r[i] = (uint64_t)acc;
    }
    uint64_t c = (uint64_t)(acc >> 64);
    memcpy(z, r, sizeof r);
    fold_in(z, (u128)c * 38);
}

static const uint64_t P25519[4] = {
    0xFFFFFFFFFFFFFFEDull, 0xFFFFFFFFFFFFFFFFull, 0xFFFFFFFFFFFFFFFFull, 0x7FFFFFFFFFFFFFFFull };

void orc_fe_mod(orc_fe x)                                   /* curve25519_mehdi.c:185-209 */
{
    for (int round = 0; round < 2; round++) {
        uint64_t r[4], b = 0;
        for (int i = 0; i < 4; i++) {
            u128 t = (u128)x[i] - P25519[i] - b;
            r[i] = (uint64_t)t;
            b = (uint64_t)(t >> 64) & 1;
        }
        uint64_t keep = (uint64_t)0 - b;          /* all-ones when x < p: keep x */
        for (int i = 0; i < 4; i++) x[i] = (x[i] & keep) | (r[i] & ~keep);
    }
}

static void sqr_n_mul(orc_fe z, const orc_fe x, int n, const orc_fe y)  /* ed25519_verify.c:108-114 */
{
    orc_fe t;
    orc_fe_sqr(t, x);
    while (--n > 0) orc_fe_sqr(t, t);
    orc_fe_mul(z, t, y);
}

/* shared front of the two addition chains: returns x^(2^250-1) and x^11 */
static void chain_250(orc_fe x250, orc_fe x11, const orc_fe x)
{
    orc_fe x2, x9, x5, x10, x20, x50, x100, t;
    orc_fe_sqr(x2, x);                 /* 2 */
    sqr_n_mul(x9, x2, 2, x);           /* 9 */
    orc_fe_mul(x11, x9, x2);           /* 11 */
    orc_fe_sqr(t, x11);                /* 22 */
    orc_fe_mul(x5, t, x9);             /* 31 = 2^5-1 */
    sqr_n_mul(x10, x5, 5, x5);         /* 2^10-1 */
    sqr_n_mul(x20, x10, 10, x10);      /* 2^20-1 */
    sqr_n_mul(t, x20, 20, x20);        /* 2^40-1 */
    sqr_n_mul(x50, t, 10, x10);        /* 2^50-1 */
    sqr_n_mul(x100, x50, 50, x50);     /* 2^100-1 */
    sqr_n_mul(t, x100, 100, x100);     /* 2^200-1 */
    sqr_n_mul(x250, t, 50, x50);       /* 2^250-1 */
}

void orc_fe_inv(orc_fe out, const orc_fe z)                 /* curve25519_mehdi.c:340-409 */
{
    orc_fe x250, x11;
    chain_250(x250, x11, z);
    sqr_n_mul(out, x250, 5, x11);      /* 2^255-32+11 = p-2 */
}

void orc_fe_pow2523(orc_fe out, const orc_fe x)             /* ed25519_verify.c:116-135 */
{
    orc_fe x250, x11, xin;
    memcpy(xin, x, sizeof xin);
    chain_250(x250, x11, xin);
    sqr_n_mul(out, x250, 2, xin);      /* 2^252-4+1 = (p-5)/8 */
}
