/*
 * oracle/orc_x25519.c -- X25519 variable-base Montgomery ladder, CPU restatement of
 * source/curve25519_dh.c (TEST INFRASTRUCTURE ONLY).
 *
 * Behaviour pinned by SURVEY.md 3.5: all 256 bits of the peer public key are used (bit 255 is NOT
 * masked, value taken mod p), no low-order check (those inputs give 32 zero bytes), sk is clamped in
 * the caller's buffer.  The reference starts from a randomised projective Z (curve25519_dh.c:123);
 * that is output-neutral, so Z = 1 here.
 */
#include "orc25519.h"
#include <string.h>

typedef struct { orc_fe X, Z; } xz_point;

/* Y = 2X   (ecp_MontDouble, curve25519_dh.c:40-54) */
static void mont_double(xz_point *Y, const xz_point *X)
{
    orc_fe A, B;
    orc_fe_add(A, X->X, X->Z);
    orc_fe_sub(B, X->X, X->Z);
    orc_fe_sqr(A, A);
    orc_fe_sqr(B, B);
    orc_fe_mul(Y->X, A, B);
    orc_fe_sub(B, A, B);
    orc_fe_mulw_add(A, A, 121665, B);
    orc_fe_mul(Y->Z, A, B);
}

/* P = P + Q, Q = 2Q, difference = (base : 1)   (ecp_Mont, curve25519_dh.c:57-84) */
static void mont_step(xz_point *P, xz_point *Q, const orc_fe base)
{
    orc_fe A, B, C, D, E;
    orc_fe_sub(A, P->X, P->Z);
    orc_fe_add(B, P->X, P->Z);
    orc_fe_sub(C, Q->X, Q->Z);
    orc_fe_add(D, Q->X, Q->Z);
    orc_fe_mul(A, A, D);
    orc_fe_mul(B, B, C);
    orc_fe_add(E, A, B);
    orc_fe_sub(B, A, B);
    orc_fe_sqr(P->X, E);
    orc_fe_sqr(A, B);
    orc_fe_mul(P->Z, A, base);

    orc_fe_sqr(A, D);
    orc_fe_sqr(B, C);
    orc_fe_mul(Q->X, A, B);
    orc_fe_sub(B, A, B);
    orc_fe_mulw_add(A, A, 121665, B);
    orc_fe_mul(Q->Z, A, B);
}

/* out = k * (pk : 1), x-only   (ecp_PointMultiply, curve25519_dh.c:94-157) */
void orc_x25519_pointmul(uint8_t out[32], const uint8_t pk[32], const uint8_t k[32])
{
    orc_fe X;
    xz_point P, Q;
    int top = 255;

    orc_fe_frombytes(X, pk);
    while (top >= 0 && !((k[top >> 3] >> (top & 7)) & 1)) top--;   /* first set bit, MSB first (:107-116) */
    if (top < 0) { memset(out, 0, 32); return; }                    /* K == 0 (:155-156) */

    memcpy(P.X, X, sizeof X);
    memset(P.Z, 0, sizeof P.Z); P.Z[0] = 1;
    mont_double(&Q, &P);                                            /* P = 1*G, Q = 2*G (:125) */

    for (int i = top - 1; i >= 0; i--) {
        int bit = (k[i >> 3] >> (i & 7)) & 1;
        /* bit=1: P = P+Q, Q = 2Q ; bit=0: Q = P+Q, P = 2P   (:89, :127-146) */
        if (bit) mont_step(&P, &Q, X); else mont_step(&Q, &P, X);
    }

    orc_fe_inv(Q.Z, P.Z);                                           /* :148 */
    orc_fe_mul(X, P.X, Q.Z);
    orc_fe_mod(X);                                                  /* ecp_MulMod :149 */
    orc_fe_tobytes(out, X);
}

void orc_x25519_shared(uint8_t shared[32], const uint8_t pk[32], uint8_t sk[32])  /* :201-208 */
{
    orc_x25519_clamp(sk);
    orc_x25519_pointmul(shared, pk, sk);
}

void orc_x25519_public(uint8_t pk[32], uint8_t sk[32])                           /* :191-198 */
{
    static const uint8_t base[32] = { 9 };
    orc_x25519_clamp(sk);
    orc_x25519_pointmul(pk, base, sk);
}

/* Edwards 8-fold walk then u = (Z+Y)/(Z-Y)   (x25519_BasePointMultiply, curve25519_dh.c:162-189) */
void orc_x25519_public_fast(uint8_t pk[32], uint8_t sk[32])
{
    orc_ext S;
    orc_fe k, u;
    orc_x25519_clamp(sk);
    orc_fe_frombytes(k, sk);
    orc_ed_basemult(&S, k);
    orc_fe_add(S.t, S.z, S.y);
    orc_fe_sub(S.z, S.z, S.y);
    orc_fe_inv(S.z, S.z);
    orc_fe_mul(u, S.t, S.z);
    orc_fe_mod(u);
    orc_fe_tobytes(pk, u);
}
