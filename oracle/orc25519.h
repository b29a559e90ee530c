/*
 * oracle/orc25519.h -- CPU restatement of the msotoodeh/curve25519 scalar-multiplication path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (curve25519_amd/, include/, the C-ABI
 * library) may include, link or call this code.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg use it, and only as the checker / the timed CPU baseline.
 *
 * Parity pin: this restatement is checked byte-for-byte against the real reference built from
 * /root/reference (oracle/_ref/libcurve25519_ref.so, recipe in oracle/Makefile) and against the
 * committed golden vectors in tests/golden/ (RFC 7748, RFC 8032, edge-case public keys,
 * seeded random batches) by tests/test_oracle_*.py.
 *
 * Written fresh: 4x64-bit limbs with unsigned __int128 (the reference's portable C uses 8x32).
 * Each function cites the reference file:line whose behaviour it restates.
 */
#ifndef ORC25519_H
#define ORC25519_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef uint64_t orc_fe[4];          /* 256-bit little-endian value, taken mod p = 2^255-19 */

typedef struct { orc_fe x, y, z, t; } orc_ext;          /* curve25519_mehdi.h:60-65  Ext_POINT */
typedef struct { orc_fe ypx, ymx, t2d; } orc_pa;        /* curve25519_mehdi.h:77-82  PA_POINT  */
typedef struct { orc_fe ypx, ymx, t2d, z2; } orc_pe;    /* curve25519_mehdi.h:68-74  PE_POINT  */
typedef struct { uint8_t pk[32]; orc_pe q[16]; } orc_sigv_ctx;  /* ed25519_verify.c:44-47 */

/* ---- field GF(2^255-19), lazy 256-bit representatives (curve25519_mehdi.c) ---- */
void orc_fe_frombytes(orc_fe y, const uint8_t *x);
void orc_fe_tobytes(uint8_t *y, const orc_fe x);
void orc_fe_add(orc_fe z, const orc_fe x, const orc_fe y);      /* ecp_AddReduce  :134 */
void orc_fe_sub(orc_fe z, const orc_fe x, const orc_fe y);      /* ecp_SubReduce  :161 */
void orc_fe_mul(orc_fe z, const orc_fe x, const orc_fe y);      /* ecp_MulReduce  :278 */
void orc_fe_sqr(orc_fe z, const orc_fe x);                      /* ecp_SqrReduce  :310 */
void orc_fe_mulw_add(orc_fe z, const orc_fe y, uint64_t b, const orc_fe x); /* ecp_WordMulAddReduce :243 */
void orc_fe_mod(orc_fe x);                                      /* ecp_Mod        :185 */
void orc_fe_inv(orc_fe out, const orc_fe z);                    /* ecp_Inverse    :340 */
void orc_fe_pow2523(orc_fe out, const orc_fe x);                /* ecp_ModExp2523 ed25519_verify.c:116 */

/* ---- scalars mod L (curve25519_order.c) ---- */
void orc_sc_from_digest(uint64_t y[4], const uint8_t md[64]);   /* eco_DigestToWords :139 (not canonical) */
void orc_sc_mod(uint64_t x[4]);                                 /* eco_Mod :125 (canonical) */
void orc_sc_mul(uint64_t z[4], const uint64_t x[4], const uint64_t y[4]);  /* eco_MulReduce :110 */
void orc_sc_add(uint64_t z[4], const uint64_t x[4], const uint64_t y[4]);  /* eco_AddReduce :132 */

/* ---- SHA-512 (sha512.c) ---- */
typedef struct { uint64_t h[8]; uint8_t buf[128]; uint64_t nbytes; } orc_sha512_ctx;
void orc_sha512_init(orc_sha512_ctx *c);
void orc_sha512_update(orc_sha512_ctx *c, const void *data, size_t n);
void orc_sha512_final(orc_sha512_ctx *c, uint8_t out[64]);

/* ---- fold recoding (curve25519_utils.c:125-153) ---- */
void orc_fold8(uint8_t cut[32], const uint64_t k[4]);
void orc_fold4(uint8_t cut[64], const uint64_t k[4]);

/* ---- Edwards (ed25519_sign.c / ed25519_verify.c) ---- */
const orc_pa *orc_base_folding8(void);        /* 256 entries, generated on first use (base_folding8.h) */
void orc_ed_double(orc_ext *p);                               /* edp_DoublePoint    ed25519_sign.c:122 */
void orc_ed_add_affine(orc_ext *p, const orc_pa *q);          /* edp_AddAffinePoint ed25519_sign.c:97  */
void orc_ed_add(orc_ext *r, const orc_ext *p, const orc_pe *q); /* edp_AddPoint     ed25519_verify.c:142 */
void orc_ed_ext2pe(orc_pe *r, const orc_ext *p);              /* edp_ExtPoint2PE    ed25519_sign.c:270 */
void orc_ed_basemult(orc_ext *s, const uint64_t k[4]);        /* edp_BasePointMult  ed25519_sign.c:215 */
void orc_ed_calc_x(orc_fe x, const orc_fe y, unsigned parity);/* ed25519_CalculateX ed25519_verify.c:66 */

/* ---- public API restatement (same byte-level behaviour as the reference's C API) ---- */
void orc_x25519_clamp(uint8_t sk[32]);                                         /* ecp_TrimSecretKey */
void orc_x25519_pointmul(uint8_t out[32], const uint8_t pk[32], const uint8_t k[32]); /* ecp_PointMultiply */
void orc_x25519_shared(uint8_t shared[32], const uint8_t pk[32], uint8_t sk[32]);  /* curve25519_dh_CreateSharedKey */
void orc_x25519_public(uint8_t pk[32], uint8_t sk[32]);                        /* curve25519_dh_CalculatePublicKey */
void orc_x25519_public_fast(uint8_t pk[32], uint8_t sk[32]);                   /* curve25519_dh_CalculatePublicKey_fast */
void orc_ed25519_keypair(uint8_t pub[32], uint8_t priv[64], const uint8_t sk[32]);   /* ed25519_CreateKeyPair */
void orc_ed25519_sign(uint8_t sig[64], const uint8_t priv[64], const uint8_t *msg, size_t n); /* ed25519_SignMessage */
void orc_ed25519_verify_init(orc_sigv_ctx *ctx, const uint8_t pk[32]);         /* ed25519_Verify_Init */
int  orc_ed25519_verify_check(const orc_sigv_ctx *ctx, const uint8_t sig[64], const uint8_t *msg, size_t n);
int  orc_ed25519_verify(const uint8_t sig[64], const uint8_t pk[32], const uint8_t *msg, size_t n);
void orc_ed25519_verify_point(uint8_t out[32], const uint8_t sig[64], const uint8_t pk[32], const uint8_t *msg, size_t n);

/* ---- batch drivers (contiguous fixed-stride arrays; nthreads<=1 runs inline) ---- */
void orc_x25519_shared_batch(uint8_t *shared, const uint8_t *pk, uint8_t *sk, size_t n, int nthreads);
void orc_x25519_public_batch(uint8_t *pk, uint8_t *sk, size_t n, int fast, int nthreads);
void orc_ed25519_keypair_batch(uint8_t *pub, uint8_t *priv, const uint8_t *sk, size_t n, int nthreads);
void orc_ed25519_sign_batch(uint8_t *sig, const uint8_t *priv, const uint8_t *msg, size_t msg_size,
                            size_t n, int nthreads);
void orc_ed25519_verify_batch(int32_t *ok, const uint8_t *sig, const uint8_t *pk, const uint8_t *msg,
                              size_t msg_size, size_t n, int nthreads);

/* the REAL reference library (oracle/_ref) driven over a thread pool: the timed CPU baseline of bench.py */
int orc_ref_x25519_shared_batch(const char *so_path, uint8_t *shared, const uint8_t *pk, uint8_t *sk, size_t n, int nthreads);
int orc_ref_ed25519_sign_batch(const char *so_path, uint8_t *sig, const uint8_t *priv, const uint8_t *msg, size_t msg_size,
                               size_t n, int nthreads);
int orc_ref_ed25519_verify_batch(const char *so_path, int32_t *ok, const uint8_t *sig, const uint8_t *pk, const uint8_t *msg,
                                 size_t msg_size, size_t n, int nthreads);

/* deterministic input generator shared by oracle, tests and bench (splitmix64 stream) */
void orc_fill_random(uint8_t *dst, size_t nbytes, uint64_t seed);

#ifdef __cplusplus
}
#endif
#endif
