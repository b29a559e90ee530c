/*
 * tests/c/dropin_test.c -- a C caller that uses ONLY the reference's public API (include/curve25519_dh.h,
 * include/ed25519_signature.h) the way the reference's own harness does (test/curve25519_test.c:
 * dh_test :429-474, signature_test :323-410 with the RFC 8032 TEST 2 vector :412-424), linked against
 * libcurve25519_amd.so instead of libcurve25519.a.  Written fresh; exit code = number of failures.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "curve25519_dh.h"
#include "ed25519_signature.h"

static int failures;

static void hex(const char *name, const unsigned char *p, size_t n)
{
    printf("%-18s", name);
    for (size_t i = 0; i < n; i++) printf("%02x", p[i]);
    printf("\n");
}

#define CHECK(cond, what) do { if (!(cond)) { failures++; printf("FAILED: %s\n", what); } else printf("ok: %s\n", what); } while (0)

static int unhex(unsigned char *out, const char *s)
{
    size_t n = strlen(s) / 2;
    for (size_t i = 0; i < n; i++) { unsigned v; sscanf(s + 2 * i, "%2x", &v); out[i] = (unsigned char)v; }
    return (int)n;
}

int main(void)
{
    /* ---- key exchange: both sides must derive the same secret; sk is clamped in place ---- */
    unsigned char a_sk[32], b_sk[32], a_pk[32], b_pk[32], a_fast[32], a_sh[32], b_sh[32], keep[32];
    unhex(a_sk, "03ac674216f3e15c761ee1a5e255f067953623c8b388b4459e13f978d7c846f4");
    unhex(b_sk, "88d4266fd4e6338d13b845fcf289579d209c897823b9217da3e161936f031589");
    memcpy(keep, a_sk, 32);
    curve25519_dh_CalculatePublicKey(a_pk, a_sk);
    curve25519_dh_CalculatePublicKey(b_pk, b_sk);
    CHECK(a_sk[0] == (keep[0] & 0xf8) && a_sk[31] == ((keep[31] | 0x40) & 0x7f), "secret key clamped in caller's buffer");
    memcpy(keep, a_sk, 32);
    curve25519_dh_CalculatePublicKey_fast(a_fast, keep);
    CHECK(memcmp(a_fast, a_pk, 32) == 0, "CalculatePublicKey_fast == CalculatePublicKey");
    curve25519_dh_CreateSharedKey(a_sh, b_pk, a_sk);
    curve25519_dh_CreateSharedKey(b_sh, a_pk, b_sk);
    hex("alice shared", a_sh, 32);
    CHECK(memcmp(a_sh, b_sh, 32) == 0, "DH shared secrets agree");
    {
        unsigned char expect[32];
        unhex(expect, "3517fe68");
        CHECK(memcmp(a_sh, expect, 4) == 0, "shared secret starts 3517fe68 (the reference's value)");
        memcpy(keep, b_pk, 32);                      /* output may alias the public key */
        curve25519_dh_CreateSharedKey(keep, keep, a_sk);
        CHECK(memcmp(keep, a_sh, 32) == 0, "shared may alias pk");
    }

    /* ---- signatures: RFC 8032 TEST 2 ---- */
    unsigned char sk[32], pk_expect[32], sig_expect[64], pub[32], priv[64], sig[64], msg[1] = { 0x72 };
    unhex(sk, "4ccd089b28ff96da9db6c346ec114e0f5b8a319f35aba624da8cf6ed4fb8a6fb");
    unhex(pk_expect, "3d4017c3e843895a92b70aa74d1b7ebc9c982ccf2ec4968cc0cd55f12af4660c");
    unhex(sig_expect, "92a009a9f0d4cab8720e820b5f642540a2b27b5416503f8fb3762223ebdb69da"
                      "085ac1e43e15996e458f3613d0f11d8c387b2eaeb4302aeeb00d291612bb0c00");
    ed25519_CreateKeyPair(pub, priv, 0, sk);
    CHECK(memcmp(pub, pk_expect, 32) == 0, "public key matches RFC 8032 TEST 2");
    CHECK(memcmp(priv, sk, 32) == 0 && memcmp(priv + 32, pub, 32) == 0, "private key = sk || pk");
    ed25519_SignMessage(sig, priv, 0, msg, sizeof msg);
    hex("signature", sig, 64);
    CHECK(memcmp(sig, sig_expect, 64) == 0, "signature matches RFC 8032 TEST 2");
    CHECK(ed25519_VerifySignature(sig, pub, msg, sizeof msg) == 1, "signature verifies");

    /* blinded path: same bytes (blinding is output-neutral) */
    {
        unsigned char seed[64], sig2[64], pub2[32], priv2[64];
        for (int i = 0; i < 64; i++) seed[i] = (unsigned char)(i * 3 + 1);
        void *bl = ed25519_Blinding_Init(0, seed, sizeof seed);
        CHECK(bl != 0, "Blinding_Init allocates");
        ed25519_CreateKeyPair(pub2, priv2, bl, sk);
        ed25519_SignMessage(sig2, priv2, bl, msg, sizeof msg);
        CHECK(memcmp(pub2, pub, 32) == 0 && memcmp(sig2, sig, 64) == 0, "blinded keygen/sign give identical bytes");
        ed25519_Blinding_Finish(bl);
    }

    /* two-phase verification, heap context and caller-provided 2080-byte context */
    {
        void *ctx = ed25519_Verify_Init(0, pub);
        CHECK(ctx != 0, "Verify_Init allocates");
        CHECK(ed25519_Verify_Check(ctx, sig, msg, sizeof msg) == 1, "Verify_Check accepts");
        sig[3] ^= 0x10;
        CHECK(ed25519_Verify_Check(ctx, sig, msg, sizeof msg) == 0, "Verify_Check rejects corrupted R");
        sig[3] ^= 0x10;
        msg[0] ^= 1;
        CHECK(ed25519_Verify_Check(ctx, sig, msg, sizeof msg) == 0, "Verify_Check rejects corrupted message");
        msg[0] ^= 1;
        ed25519_Verify_Finish(ctx);
        unsigned char storage[2080];
        CHECK(ed25519_Verify_Init(storage, pub) == (void *)storage, "Verify_Init uses caller storage");
        CHECK(ed25519_Verify_Check(storage, sig, msg, sizeof msg) == 1, "Verify_Check with caller storage");
    }
    printf("%d failure(s)\n", failures);
    return failures;
}
