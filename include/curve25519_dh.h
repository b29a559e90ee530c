/*
 * include/curve25519_dh.h -- X25519 key agreement, drop-in for msotoodeh/curve25519's
 * include/curve25519_dh.h (the three prototypes at reference include/curve25519_dh.h:34-48).
 *
 * Same names, same argument order, same byte-level behaviour as the reference's portable-C build:
 *   - every buffer is 32 bytes, little-endian;
 *   - `sk` is IN/OUT: it is clamped in the caller's buffer (reference source/curve25519_dh.c:186,196,206);
 *   - all 256 bits of a peer public key are used (bit 255 is not masked), nothing is validated,
 *     low-order inputs yield 32 zero bytes;
 *   - the functions return void: there is no error channel.  This library computes on an AMD
 *     MI355X (gfx950) through HIP and ABORTS the process with a message on stderr if no usable
 *     device is present -- it never falls back to a CPU implementation.
 *
 * A single call is a device batch of one.  Calls of up to a few thousand elements run ONE operation per wave
 * (csrc/coop25519.cuh: a field element limb-per-lane, four field products at a time; curve25519_dh_CreateSharedKey: two waves
 * per element, a ladder step in two product levels), the inversion by division steps on a quad of lanes, and a call of one
 * returns on a completion word its kernel stores behind the result: 0.139 ms for curve25519_dh_CreateSharedKey, 0.12 ms for
 * curve25519_dh_CalculatePublicKey, 0.04 ms for _fast, end to end (profiles/r06_single_call.txt; round 5: 0.17 / 0.16 /
 * 0.07; round 3's one-operation-per-lane pass: 0.76 ms) against 93 / 93 / 43 us for the reference on one host core of the
 * same box -- the literal drop-in call is correct and five times faster than it was; only _fast is faster than a host core.
 * Throughput comes from the batch entry points in curve25519_amd.h: a call of 2 operations already beats one host core, ~40
 * beat sixteen cores (a call of 1024 takes 0.19 ms, one of 2^14 -- four lanes per element, csrc/quad25519.cuh -- 0.33 ms =
 * 50 M/s: profiles/r06_small_batch_sweep.txt, r06_mid_batch_sweep.txt); the quoted rates need 2^17 and more per call
 * (2^17: 117 M/s, 2^20: 131-134 M/s).
 */
#ifndef CURVE25519_AMD_DH_H
#define CURVE25519_AMD_DH_H

#ifdef __cplusplus
extern "C" {
#endif

/* pk = clamp(sk) * 9 by the Montgomery ladder.   replaces reference include/curve25519_dh.h:34 */
void curve25519_dh_CalculatePublicKey(
    unsigned char *pk,          /* [32 bytes] OUT: public key */
    unsigned char *sk);         /* [32 bytes] IN/OUT: secret key, clamped on return */

/* Same result through the Edwards 8-fold fixed-base walk.   replaces reference :40 */
void curve25519_dh_CalculatePublicKey_fast(
    unsigned char *pk,          /* [32 bytes] OUT: public key */
    unsigned char *sk);         /* [32 bytes] IN/OUT: secret key, clamped on return */

/* shared = clamp(sk) * pk.   replaces reference :45.  `shared` may alias `pk`. */
void curve25519_dh_CreateSharedKey(
    unsigned char *shared,      /* [32 bytes] OUT: shared secret */
    const unsigned char *pk,    /* [32 bytes] IN: peer public key */
    unsigned char *sk);         /* [32 bytes] IN/OUT: secret key, clamped on return */

#ifdef __cplusplus
}
#endif
#endif
