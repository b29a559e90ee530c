/*
 * oracle/orc_ed25519.c -- Ed25519 keygen / sign / verify with the reference's 8-fold and 4-fold
 * fixed-base walks, CPU restatement of source/ed25519_sign.c and source/ed25519_verify.c
 * (TEST INFRASTRUCTURE ONLY).
 *
 * Curve constants (d, 2d, 1/d, sqrt(-1), the base point) and the 256-entry folding table are
 * derived at first use from their definitions (d = -121665/121666, B.y = 4/5, table entry k =
 * sum over set bits i of k of 2^(32i)*B -- the recipe of test/curve25519_selftest.c:498-551), not
 * copied; tests compare the table with the reference's source/base_folding8.h via oracle/_ref.
 *
 * Behaviour pinned by SURVEY.md 3.5: verify does NOT check S < L, does not validate the public key,
 * compares only enc(R) bytes; the projective randomiser R of edp_BasePointMult is output-neutral
 * (R = 1 here).
 */
#include "orc25519.h"
#include <pthread.h>
#include <string.h>

static orc_fe C_d, C_2d, C_di, C_I;
static const orc_fe C_one = {1, 0, 0, 0}, C_zero = {0, 0, 0, 0};
/* 2p = 2^256-38, used for negation (curve25519_mehdi.c:56-59) */
static const orc_fe C_maxP = {0xFFFFFFFFFFFFFFDAull, ~0ull, ~0ull, ~0ull};
static orc_pa TBL8[256];
static pthread_once_t once = PTHREAD_ONCE_INIT;

static void fe_set(orc_fe x, uint64_t v) { x[0] = v; x[1] = x[2] = x[3] = 0; }

void orc_ed_double(orc_ext *p)                          /* ed25519_sign.c:122-143 */
{
    orc_fe a, b, c, d, e;
    orc_fe_sqr(a, p->x);
    orc_fe_sqr(b, p->y);
    orc_fe_sqr(c, p->z);
    orc_fe_add(c, c, c);
    orc_fe_sub(d, C_maxP, a);           /* D = -A = 2p - A (:130) */
    orc_fe_sub(a, d, b);                /* H = D-B */
    orc_fe_add(d, d, b);                /* G = D+B */
    orc_fe_sub(b, d, c);                /* F = G-C */
    orc_fe_add(e, p->x, p->y);
    orc_fe_sqr(e, e);
    orc_fe_add(e, e, a);                /* E = (X+Y)^2 + H */
    orc_fe_mul(p->x, e, b);
    orc_fe_mul(p->y, a, d);
    orc_fe_mul(p->z, d, b);
    orc_fe_mul(p->t, e, a);
}

void orc_ed_add_affine(orc_ext *p, const orc_pa *q)     /* ed25519_sign.c:97-115 */
{
    orc_fe a, b, c, d, e;
    orc_fe_sub(a, p->y, p->x);
    orc_fe_mul(a, a, q->ymx);
    orc_fe_add(b, p->y, p->x);
    orc_fe_mul(b, b, q->ypx);
    orc_fe_mul(c, p->t, q->t2d);
    orc_fe_add(d, p->z, p->z);
    orc_fe_sub(e, b, a);
    orc_fe_add(b, b, a);
    orc_fe_sub(a, d, c);
    orc_fe_add(d, d, c);
    orc_fe_mul(p->x, e, a);
    orc_fe_mul(p->y, b, d);
    orc_fe_mul(p->t, e, b);
    orc_fe_mul(p->z, d, a);
}

void orc_ed_add(orc_ext *r, const orc_ext *p, const orc_pe *q)   /* ed25519_verify.c:142-161 */
{
    orc_fe a, b, c, d, e;
    orc_fe_sub(a, p->y, p->x);
    orc_fe_mul(a, a, q->ymx);
    orc_fe_add(b, p->y, p->x);
    orc_fe_mul(b, b, q->ypx);
    orc_fe_mul(c, p->t, q->t2d);
    orc_fe_mul(d, p->z, q->z2);
    orc_fe_sub(e, b, a);
    orc_fe_add(b, b, a);
    orc_fe_sub(a, d, c);
    orc_fe_add(d, d, c);
    orc_fe_mul(r->x, e, a);
    orc_fe_mul(r->y, b, d);
    orc_fe_mul(r->t, e, b);
    orc_fe_mul(r->z, d, a);
}

void orc_ed_ext2pe(orc_pe *r, const orc_ext *p)         /* ed25519_sign.c:270-276 */
{
    orc_fe_add(r->ypx, p->y, p->x);
    orc_fe_sub(r->ymx, p->y, p->x);
    orc_fe_mul(r->t2d, p->t, C_2d);
    orc_fe_add(r->z2, p->z, p->z);
}

/* x from y: sqrt((y^2-1)/(d y^2+1)), no on-curve rejection   (ed25519_verify.c:66-100) */
static void calc_x(orc_fe X, const orc_fe Y, unsigned parity)
{
    orc_fe u, v, a, b;
    orc_fe_sqr(u, Y);
    orc_fe_mul(v, u, C_d);
    orc_fe_sub(u, u, C_one);
    orc_fe_add(v, v, C_one);

    orc_fe_sqr(b, v);
    orc_fe_mul(a, u, b);
    orc_fe_mul(a, a, v);                /* a = u v^3 */
    orc_fe_sqr(b, b);
    orc_fe_mul(b, a, b);                /* b = u v^7 */
    orc_fe_pow2523(b, b);
    orc_fe_mul(X, b, a);

    orc_fe_sqr(b, X);
    orc_fe_mul(b, b, v);
    orc_fe_sub(b, b, u);
    orc_fe_mod(b);
    if (b[0] | b[1] | b[2] | b[3]) orc_fe_mul(X, X, C_I);    /* :92-93 */

    orc_fe_mod(X);                                           /* :95 */
    if ((X[0] ^ parity) & 1) {                               /* :98-99  X = p - X */
        static const uint64_t P[4] = {
            0xFFFFFFFFFFFFFFEDull, 0xFFFFFFFFFFFFFFFFull, 0xFFFFFFFFFFFFFFFFull, 0x7FFFFFFFFFFFFFFFull };
        uint64_t bw = 0;
        for (int i = 0; i < 4; i++) {
            unsigned __int128 t = (unsigned __int128)P[i] - X[i] - bw;
            X[i] = (uint64_t)t;
            bw = (uint64_t)(t >> 64) & 1;
        }
    }
}

static void init_constants(void)
{
    orc_fe t, n;

    /* d = -121665/121666 */
    fe_set(t, 121666);
    orc_fe_inv(t, t);
    fe_set(n, 121665);
    orc_fe_sub(n, C_zero, n);
    orc_fe_mul(C_d, n, t);  orc_fe_mod(C_d);
    orc_fe_add(C_2d, C_d, C_d); orc_fe_mod(C_2d);
    orc_fe_inv(C_di, C_d);  orc_fe_mod(C_di);

    /* sqrt(-1) = 2^((p-1)/4) = (2^((p-5)/8))^2 * 2 */
    fe_set(t, 2);
    orc_fe_pow2523(n, t);
    orc_fe_sqr(n, n);
    orc_fe_mul(C_I, n, t);  orc_fe_mod(C_I);

    /* base point: y = 4/5, x even */
    orc_ext P[8];
    orc_fe by, bx;
    fe_set(t, 5);
    orc_fe_inv(t, t);
    fe_set(n, 4);
    orc_fe_mul(by, n, t); orc_fe_mod(by);
    calc_x(bx, by, 0);

    memcpy(P[0].x, bx, sizeof bx);
    memcpy(P[0].y, by, sizeof by);
    fe_set(P[0].z, 1);
    orc_fe_mul(P[0].t, bx, by);
    for (int i = 1; i < 8; i++) {                       /* P[i] = 2^(32 i) * B */
        P[i] = P[i - 1];
        for (int j = 0; j < 32; j++) orc_ed_double(&P[i]);
    }

    /* E[k] = sum of P[i] over set bits of k, then to canonical affine (Y+X, Y-X, 2dT) */
    static orc_ext E[256];
    fe_set(E[0].x, 0); fe_set(E[0].y, 1); fe_set(E[0].z, 1); fe_set(E[0].t, 0);
    for (int k = 1; k < 256; k++) {
        int hi = 7;
        while (!((k >> hi) & 1)) hi--;
        orc_pe q;
        orc_ed_ext2pe(&q, &P[hi]);
        orc_ed_add(&E[k], &E[k ^ (1 << hi)], &q);
    }
    for (int k = 0; k < 256; k++) {
        orc_fe zi, x, y;
        orc_fe_inv(zi, E[k].z);
        orc_fe_mul(x, E[k].x, zi);
        orc_fe_mul(y, E[k].y, zi);
        orc_fe_add(TBL8[k].ypx, y, x);  orc_fe_mod(TBL8[k].ypx);
        orc_fe_sub(TBL8[k].ymx, y, x);  orc_fe_mod(TBL8[k].ymx);
        orc_fe_mul(t, x, y);
        orc_fe_mul(TBL8[k].t2d, t, C_2d); orc_fe_mod(TBL8[k].t2d);
    }
}

const orc_pa *orc_base_folding8(void)
{
    pthread_once(&once, init_constants);
    return TBL8;
}

void orc_ed_calc_x(orc_fe x, const orc_fe y, unsigned parity)
{
    pthread_once(&once, init_constants);
    calc_x(x, y, parity);
}

/* S = k*B by the 8-fold walk, projective result   (edp_BasePointMult, ed25519_sign.c:215-244) */
void orc_ed_basemult(orc_ext *S, const uint64_t k[4])
{
    uint8_t cut[32];
    pthread_once(&once, init_constants);
    orc_fold8(cut, k);

    const orc_pa *p0 = &TBL8[cut[0]];
    orc_fe_sub(S->x, p0->ypx, p0->ymx);        /* 2x */
    orc_fe_add(S->y, p0->ypx, p0->ymx);        /* 2y */
    orc_fe_mul(S->t, p0->t2d, C_di);           /* 2xy */
    fe_set(S->z, 2);                           /* Z = 2R with R = 1 */
    for (int i = 1; i < 32; i++) {
        orc_ed_double(S);
        orc_ed_add_affine(S, &TBL8[cut[i]]);
    }
}

/* affine canonical (x, y) = k*B   (edp_BasePointMultiply, ed25519_sign.c:246-268, blinding == NULL) */
static void basemult_affine(orc_fe x, orc_fe y, const uint64_t k[4])
{
    orc_ext S;
    orc_ed_basemult(&S, k);
    orc_fe_inv(S.z, S.z);
    orc_fe_mul(x, S.x, S.z); orc_fe_mod(x);
    orc_fe_mul(y, S.y, S.z); orc_fe_mod(y);
}

/* ecp_EncodeInt, curve25519_utils.c:77-98: y with the parity of x in bit 255 */
static void pack_point(uint8_t out[32], const orc_fe y, uint64_t x0)
{
    orc_fe_tobytes(out, y);
    out[31] = (uint8_t)((out[31] & 0x7f) | ((x0 & 1) << 7));
}

void orc_ed25519_keypair(uint8_t pub[32], uint8_t priv[64], const uint8_t sk[32])   /* ed25519_sign.c:344-367 */
{
    uint8_t md[64];
    orc_sha512_ctx H;
    orc_fe a, x, y;
    orc_sha512_init(&H);
    orc_sha512_update(&H, sk, 32);
    orc_sha512_final(&H, md);
    orc_x25519_clamp(md);
    orc_fe_frombytes(a, md);
    basemult_affine(x, y, a);
    pack_point(pub, y, x[0]);
    memmove(priv, sk, 32);
    memcpy(priv + 32, pub, 32);
}

void orc_ed25519_sign(uint8_t sig[64], const uint8_t priv[64], const uint8_t *msg, size_t n)  /* :372-419 */
{
    uint8_t md[64], rs[64];
    orc_sha512_ctx H;
    uint64_t a[4], r[4], t[4];
    orc_fe x, y;

    orc_sha512_init(&H);
    orc_sha512_update(&H, priv, 32);
    orc_sha512_final(&H, md);
    orc_x25519_clamp(md);
    orc_fe_frombytes(a, md);

    orc_sha512_init(&H);                          /* r = H(b || m) mod L */
    orc_sha512_update(&H, md + 32, 32);
    orc_sha512_update(&H, msg, n);
    orc_sha512_final(&H, md);
    orc_sc_from_digest(r, md);
    orc_sc_mod(r);

    basemult_affine(x, y, r);                     /* R = r*B */
    pack_point(rs, y, x[0]);

    orc_sha512_init(&H);                          /* h = H(enc(R) || pk || m) */
    orc_sha512_update(&H, rs, 32);
    orc_sha512_update(&H, priv + 32, 32);
    orc_sha512_update(&H, msg, n);
    orc_sha512_final(&H, md);
    orc_sc_from_digest(t, md);

    orc_sc_mul(t, t, a);                          /* S = h*a + r mod L */
    orc_sc_add(t, t, r);
    orc_sc_mod(t);
    orc_fe_tobytes(rs + 32, t);
    memcpy(sig, rs, 64);
}

void orc_ed25519_verify_init(orc_sigv_ctx *ctx, const uint8_t pk[32])   /* ed25519_verify.c:179-232 */
{
    orc_ext Q, T;
    uint8_t yb[32];
    int i;
    pthread_once(&once, init_constants);

    memcpy(ctx->pk, pk, 32);
    memcpy(yb, pk, 32);
    unsigned parity = yb[31] >> 7;                /* ecp_DecodeInt curve25519_utils.c:100-123 */
    yb[31] &= 0x7f;
    orc_fe_frombytes(Q.y, yb);
    calc_x(Q.x, Q.y, ~parity);                    /* inverted parity: Q = -A (:193) */
    orc_fe_mul(Q.t, Q.x, Q.y); orc_fe_mod(Q.t);
    fe_set(Q.z, 1);

    fe_set(ctx->q[0].ypx, 1); fe_set(ctx->q[0].ymx, 1); fe_set(ctx->q[0].t2d, 0); fe_set(ctx->q[0].z2, 2);
    orc_ed_ext2pe(&ctx->q[1], &Q);
#define QSET(d, s) do { orc_ed_add(&T, &Q, &ctx->q[s]); orc_ed_ext2pe(&ctx->q[d], &T); } while (0)
    for (i = 0; i < 64; i++) orc_ed_double(&Q);
    orc_ed_ext2pe(&ctx->q[2], &Q);
    QSET(3, 1);
    for (; i < 128; i++) orc_ed_double(&Q);
    orc_ed_ext2pe(&ctx->q[4], &Q);
    QSET(5, 1); QSET(6, 2); QSET(7, 3);
    for (; i < 192; i++) orc_ed_double(&Q);
    orc_ed_ext2pe(&ctx->q[8], &Q);
    QSET(9, 1); QSET(10, 2); QSET(11, 3); QSET(12, 4); QSET(13, 5); QSET(14, 6); QSET(15, 7);
#undef QSET
}

/* T = s*B + h*(-A): 4-fold over the per-key table, 8-fold over the base table (ed25519_verify.c:243-280) */
static void poly_mult(orc_fe x, orc_fe y, const uint64_t s[4], const uint64_t h[4], const orc_pe *qt)
{
    uint8_t u[32], v[64];
    orc_ext S;
    int i;
    orc_fold8(u, s);
    orc_fold4(v, h);

    const orc_pe *q0 = &qt[v[0]];
    orc_fe_sub(S.x, q0->ypx, q0->ymx);
    orc_fe_add(S.y, q0->ypx, q0->ymx);
    orc_fe_mul(S.t, q0->t2d, C_di);
    memcpy(S.z, q0->z2, sizeof S.z);

    for (i = 1; i < 32; i++) {
        orc_ed_double(&S);
        orc_ed_add(&S, &S, &qt[v[i]]);
    }
    for (; i < 64; i++) {
        orc_ed_double(&S);
        orc_ed_add_affine(&S, &TBL8[u[i - 32]]);
        orc_ed_add(&S, &S, &qt[v[i]]);
    }
    orc_fe_inv(S.z, S.z);
    orc_fe_mul(x, S.x, S.z); orc_fe_mod(x);
    orc_fe_mul(y, S.y, S.z); orc_fe_mod(y);
}

int orc_ed25519_verify_check(const orc_sigv_ctx *ctx, const uint8_t sig[64], const uint8_t *msg, size_t n)
{                                                                     /* ed25519_verify.c:287-313 */
    orc_sha512_ctx H;
    uint8_t md[64];
    uint64_t h[4], s[4];
    orc_fe x, y;

    orc_sha512_init(&H);
    orc_sha512_update(&H, sig, 32);
    orc_sha512_update(&H, ctx->pk, 32);
    orc_sha512_update(&H, msg, n);
    orc_sha512_final(&H, md);
    orc_sc_from_digest(h, md);
    orc_sc_mod(h);

    orc_fe_frombytes(s, sig + 32);                /* raw 256 bits, no S < L check (:308) */
    poly_mult(x, y, s, h, ctx->q);
    pack_point(md, y, x[0]);
    return memcmp(md, sig, 32) == 0 ? 1 : 0;
}

/* test hook: the packed point T = s*B + h*(-A) that ed25519_Verify_Check compares with enc(R) (:309-310) */
void orc_ed25519_verify_point(uint8_t out[32], const uint8_t sig[64], const uint8_t pk[32], const uint8_t *msg, size_t n)
{
    orc_sigv_ctx ctx;
    orc_sha512_ctx H;
    uint8_t md[64];
    uint64_t h[4], s[4];
    orc_fe x, y;
    orc_ed25519_verify_init(&ctx, pk);
    orc_sha512_init(&H);
    orc_sha512_update(&H, sig, 32);
    orc_sha512_update(&H, ctx.pk, 32);
    orc_sha512_update(&H, msg, n);
    orc_sha512_final(&H, md);
    orc_sc_from_digest(h, md);
    orc_sc_mod(h);
    orc_fe_frombytes(s, sig + 32);
    poly_mult(x, y, s, h, ctx.q);
    pack_point(out, y, x[0]);
}

int orc_ed25519_verify(const uint8_t sig[64], const uint8_t pk[32], const uint8_t *msg, size_t n)  /* :163-173 */
{
    orc_sigv_ctx ctx;
    orc_ed25519_verify_init(&ctx, pk);
    return orc_ed25519_verify_check(&ctx, sig, msg, n);
}
