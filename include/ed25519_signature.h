/*
 * include/ed25519_signature.h -- Ed25519 sign / verify, drop-in for msotoodeh/curve25519's
 * include/ed25519_signature.h (size macros :34-37, prototypes :40-93).
 *
 * Same names, argument order and byte-level behaviour as the reference's portable-C build:
 * public key 32 B (y with x parity in bit 255), private key 64 B = secret || public,
 * signature 64 B = enc(R) || S.  Verification is the reference's: S is NOT required to be < L,
 * the public key is not validated, only the enc(R) bytes are compared (reference
 * source/ed25519_verify.c:287-313).  All arithmetic runs on an AMD MI355X (gfx950) through HIP;
 * a call made with no usable device aborts the process -- there is no CPU fallback.
 *
 * Blinding is real: ed25519_Blinding_Init derives the reference's 192-byte context (bl, zr, BP) on the
 * device, and a non-NULL context makes key generation / signing walk the scalar k + bl from a starting
 * point whose Z is randomised by zr and add BP afterwards (reference source/ed25519_sign.c:254-259,
 * :289-331).  It is output-neutral, in the reference and here: signatures and public keys are
 * byte-identical with and without a context.
 *
 * Cost of ONE call: these are the reference's single-operation prototypes, and each call runs as a device batch of
 * one.  Calls of up to 1024-2048 elements run ONE operation per wave (csrc/coop25519.cuh; verification: one launch of three
 * waves per signature -- hashing and the two square roots side by side, then the three scalar products side by side), the
 * inversion by division steps on a quad of lanes, and a call of one returns on a completion word its last kernel stores
 * behind the results: 43 us per ed25519_CreateKeyPair, 62 us per ed25519_SignMessage, 127 us per ed25519_VerifySignature, 128 /
 * 103 us per ed25519_Verify_Init / _Check, end to end (profiles/r06_single_call.txt; round 5: 76 / 105 / 135, 131 / 133;
 * round 3: 0.18 / 0.21 / 0.65 ms).  The reference on one host core of the same box: 43 us per key pair and per signature,
 * 190 us per verification -- a single verification is faster here, a single signature is not.
 * The device pays off through the *_batch / *_dev forms in curve25519_amd.h: from 2 signatures per call a batch beats one
 * host core (1024 signatures take 81 us, 2^14 -- four lanes per element, csrc/quad25519.cuh -- 77 us = 212 M/s:
 * profiles/r06_small_batch_sweep.txt, r06_mid_batch_sweep.txt), from a few dozen it beats sixteen, and the quoted
 * throughput needs 2^17 and more per call.
 */
#ifndef CURVE25519_AMD_ED25519_SIGNATURE_H
#define CURVE25519_AMD_ED25519_SIGNATURE_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ed25519_public_key_size     32
#define ed25519_secret_key_size     32
#define ed25519_private_key_size    64
#define ed25519_signature_size      64

/* replaces reference include/ed25519_signature.h:40 */
void ed25519_CreateKeyPair(
    unsigned char *pubKey,              /* OUT: [32 bytes] public key */
    unsigned char *privKey,             /* OUT: [64 bytes] private key (sk || pk) */
    const void *blinding,               /* IN: NULL or a context from ed25519_Blinding_Init */
    const unsigned char *sk);           /* IN: [32 bytes] secret key */

/* replaces reference :47 */
void ed25519_SignMessage(
    unsigned char *signature,           /* OUT: [64 bytes] signature (R, S) */
    const unsigned char *privKey,       /* IN: [64 bytes] private key (sk || pk) */
    const void *blinding,               /* IN: NULL or a blinding context */
    const unsigned char *msg,           /* IN: [msg_size bytes] message */
    size_t msg_size);

/* replaces reference :54.  context == NULL allocates (free with ed25519_Blinding_Finish);
 * otherwise the caller's storage (at least 192 bytes, the reference's EDP_BLINDING_CTX) is used. */
void *ed25519_Blinding_Init(
    void *context,
    const unsigned char *seed,          /* IN: [size bytes] blinding seed */
    size_t size);

/* replaces reference :59 */
void ed25519_Blinding_Finish(void *context);

/* One-shot verification: 1 = valid, 0 = invalid.   replaces reference :67 */
int ed25519_VerifySignature(
    const unsigned char *signature,     /* IN: [64 bytes] signature (R, S) */
    const unsigned char *publicKey,     /* IN: [32 bytes] public key */
    const unsigned char *msg,           /* IN: [msg_size bytes] message */
    size_t msg_size);

/* Two-phase verification, first part: per-key precomputation.   replaces reference :77.
 * context == NULL allocates (free with ed25519_Verify_Finish); otherwise the caller's storage
 * (at least 2080 bytes, the reference's EDP_SIGV_CTX) is filled.  Returns NULL on allocation failure. */
void *ed25519_Verify_Init(
    void *context,
    const unsigned char *publicKey);    /* IN: [32 bytes] public key */

/* Second part: check one (signature, message) pair against the context.   replaces reference :86 */
int ed25519_Verify_Check(
    const void *context,
    const unsigned char *signature,     /* IN: [64 bytes] signature (R, S) */
    const unsigned char *msg,
    size_t msg_size);

/* replaces reference :93 (unconditional free, like the reference) */
void ed25519_Verify_Finish(void *ctx);

#ifdef __cplusplus
}
#endif
#endif
