/*
 * include/curve25519_amd.h -- batched entry points of the MI355X engine (C ABI).
 *
 * The reference has no batch API: its callers loop over the single-call functions of
 * include/curve25519_dh.h and include/ed25519_signature.h (e.g. test/curve25519_test.c:144-318).
 * Each function below is the N-element form of exactly one reference call and produces, element by
 * element, the bytes that call would produce.  Layouts are contiguous fixed-stride arrays:
 *
 *     sk, pk, shared : n x 32 bytes          priv, sig : n x 64 bytes
 *     msg            : n x msg_size bytes    verdict   : n x int (1 valid / 0 invalid)
 *
 * Two flavours:
 *   *_batch : host pointers.  Synchronous: uploads, runs the kernels, downloads.  Callable from
 *             several host threads at once (each thread owns its stream and staging buffers).
 *   *_dev   : device pointers (hipMalloc'ed memory of the current device, 16-byte aligned) and a
 *             hipStream_t passed as void* (NULL = default stream).  Asynchronous: returns after
 *             enqueueing.  This is what bench.py times with inputs resident in HBM.
 *
 * Return value: 0 on success, otherwise a HIP error code (c25519_amd_last_error() gives the text).
 * There is no CPU fallback: without a usable gfx950 device every entry point fails.
 */
#ifndef CURVE25519_AMD_H
#define CURVE25519_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* library / device status ------------------------------------------------------------------- */
const char *c25519_amd_version(void);
const char *c25519_amd_last_error(void);               /* per-thread, "" when none */
int  c25519_amd_device_count(void);                    /* usable HIP devices (0 when none) */
int  c25519_amd_set_device(int device);                /* device used by this host thread */

/* X25519 ------------------------------------------------------------------------------------ */
/* n x curve25519_dh_CreateSharedKey (reference include/curve25519_dh.h:45); sk is clamped in place */
int curve25519_dh_CreateSharedKey_batch(unsigned char *shared, const unsigned char *pk,
                                        unsigned char *sk, size_t n);
int curve25519_dh_CreateSharedKey_dev(void *shared, const void *pk, void *sk, size_t n, void *stream);

/* n x curve25519_dh_CalculatePublicKey (reference :34): ladder on the base point u = 9 */
int curve25519_dh_CalculatePublicKey_batch(unsigned char *pk, unsigned char *sk, size_t n);
int curve25519_dh_CalculatePublicKey_dev(void *pk, void *sk, size_t n, void *stream);

/* n x curve25519_dh_CalculatePublicKey_fast (reference :40): Edwards 8-fold walk + birational map */
int curve25519_dh_CalculatePublicKey_fast_batch(unsigned char *pk, unsigned char *sk, size_t n);
int curve25519_dh_CalculatePublicKey_fast_dev(void *pk, void *sk, size_t n, void *stream);

/* Ed25519 ----------------------------------------------------------------------------------- */
/* n x ed25519_CreateKeyPair (reference include/ed25519_signature.h:40), blinding = NULL */
int ed25519_CreateKeyPair_batch(unsigned char *pub, unsigned char *priv, const unsigned char *sk, size_t n);
int ed25519_CreateKeyPair_dev(void *pub, void *priv, const void *sk, size_t n, void *stream);

/* n x ed25519_SignMessage (reference :47), blinding = NULL, all messages msg_size bytes long */
int ed25519_SignMessage_batch(unsigned char *sig, const unsigned char *priv, const unsigned char *msg,
                              size_t msg_size, size_t n);
int ed25519_SignMessage_dev(void *sig, const void *priv, const void *msg, size_t msg_size, size_t n,
                            void *stream);

/* the same with messages of different lengths: message i is msgs[offsets[i] .. offsets[i+1]),
 * offsets has n+1 entries (host memory for _batch, device memory for _dev) */
int ed25519_SignMessage_ragged_batch(unsigned char *sig, const unsigned char *priv, const unsigned char *msgs,
                                     const uint64_t *offsets, size_t n);
int ed25519_SignMessage_ragged_dev(void *sig, const void *priv, const void *msgs, const uint64_t *offsets,
                                   size_t n, void *stream);

/* n x ed25519_VerifySignature (reference :67): full Init + Check per element, distinct keys */
int ed25519_VerifySignature_batch(int *verdict, const unsigned char *sig, const unsigned char *pk,
                                  const unsigned char *msg, size_t msg_size, size_t n);
int ed25519_VerifySignature_dev(void *verdict, const void *sig, const void *pk, const void *msg,
                                size_t msg_size, size_t n, void *stream);

int ed25519_VerifySignature_ragged_batch(int *verdict, const unsigned char *sig, const unsigned char *pk,
                                         const unsigned char *msgs, const uint64_t *offsets, size_t n);
int ed25519_VerifySignature_ragged_dev(void *verdict, const void *sig, const void *pk, const void *msgs,
                                       const uint64_t *offsets, size_t n, void *stream);

/* bytes of device scratch ed25519_VerifySignature_dev needs for n elements (per-lane 4-fold tables);
 * the library allocates and caches it per host thread (2560 bytes per element). */
size_t ed25519_VerifySignature_scratch_bytes(size_t n);

/* Two-phase verification (reference include/ed25519_signature.h:77-93): one key, many signatures.
 * A context is 2080 bytes -- the reference's EDP_SIGV_CTX size: the 32-byte key followed by the 16-row
 * 4-fold table of 2^(64i)*(-A) subset sums, four canonical 32-byte field elements per row.
 *   Verify_Init_batch / _dev : n keys -> n contexts (n x 2080 bytes)
 *   Verify_Check_batch / _dev: ONE context, n (signature, message) pairs -> n verdicts; this is the
 *                              amortised path, the per-key table is staged in LDS. */
int ed25519_Verify_Init_batch(void *ctx, const unsigned char *pk, size_t n);
int ed25519_Verify_Init_dev(void *ctx, const void *pk, size_t n, void *stream);
int ed25519_Verify_Check_batch(int *verdict, const void *ctx, const unsigned char *sig,
                               const unsigned char *msg, size_t msg_size, size_t n);
int ed25519_Verify_Check_dev(void *verdict, const void *ctx, const void *sig, const void *msg,
                             size_t msg_size, size_t n, void *stream);

/* introspection used by tests and bench ------------------------------------------------------ */
/* copies the device-generated 256 x 96-byte 8-fold base table (canonical Y+X, Y-X, 2dT rows --
 * the content of reference source/base_folding8.h) to `out` */
int c25519_amd_base_table(unsigned char *out /* 24576 bytes */);

/* test hook: the encoded point T = s*B + h*(-A) that ed25519_Verify_Check compares with enc(R)
 * (reference source/ed25519_verify.c:309-310) instead of the verdict; device pointers, out is n x 32 bytes */
int c25519_amd_verify_point_dev(void *out, const void *sig, const void *pk, const void *msg, size_t msg_size,
                                size_t n, void *stream);

/* device field arithmetic on n pairs of 32-byte little-endian values taken mod p = 2^255-19 (host pointers):
 * out[i] = canonical(op(a[i], b[i])), op 0 mul, 1 square, 2 add, 3 sub, 4 inverse, 5 a^((p-5)/8),
 * 6 canonicalise, 7 (a-b)*(a+b).  The unit-test hook for the L0 layer (the reference's ECP_SELF_TEST checks). */
int c25519_amd_fe_selftest(unsigned char *out, const unsigned char *a, const unsigned char *b, size_t n, int op);

#ifdef __cplusplus
}
#endif
#endif
