// tests/c/cxx_wrappers_test.cpp -- SURVEY.md 8(f3): the reference's C++ wrapper classes (C++/x25519.cpp,
// C++/ed25519.cpp, compiled where they lie by `make -C oracle ref-cxx`) running on top of libcurve25519_amd.so.
// X25519Private::CreateSharedKey = DH + SHA-512 KDF (C++/x25519.cpp:72-95); ED25519Private signs with the reference's
// STATIC blinding contexts (C++/custom_blinds.h, used at C++/ed25519.cpp:87,:124), so this also checks that contexts
// produced by the reference's own tooling drive this library's blinded walk to the right bytes.
// Expected values: RFC 7748 section 6.1 and RFC 8032 section 7.1 (TEST 1-3).  Exit code = number of failures.
#include <cstdio>
#include <cstring>
#include <string>

#include "C++/ed25519.h"
#include "C++/x25519.h"
extern "C" {
#include "include/curve25519_dh.h"
#include "source/sha512.h"
}

static int failures = 0;

static std::string hex(const unsigned char* p, size_t n)
{
    static const char* d = "0123456789abcdef";
    std::string s;
    for (size_t i = 0; i < n; i++) { s += d[p[i] >> 4]; s += d[p[i] & 15]; }
    return s;
}
static void unhex(unsigned char* out, const char* s)
{
    for (size_t i = 0; s[2 * i]; i++) {
        unsigned v;
        sscanf(s + 2 * i, "%2x", &v);
        out[i] = (unsigned char)v;
    }
}
static void check(bool ok, const char* what)
{
    printf("%s  %s\n", ok ? "ok  " : "FAIL", what);
    if (!ok) failures++;
}

int main()
{
    // ---- X25519Private: RFC 7748 6.1 ----
    unsigned char a_sk[32], b_sk[32], pk[32], sh1[32], sh2[32], k1[64], k2[64];
    unhex(a_sk, "77076d0a7318a57d3c16c17251b26645df4c2f87ebc0992ab177fba51db92c2a");
    unhex(b_sk, "5dab087e624a8a4b79e17f8b83800ee66f3bb1292618b6fd1c2f8b27ff88e0eb");
    X25519Private alice(a_sk), bob(b_sk);
    check(hex(alice.GetPublicKey(pk), 32) == "8520f0098930a754748b7ddcb43ef75a0dbf3a0d26381af4eba4a98eaa9b4e6a", "alice public key (CalculatePublicKey_fast)");
    check(hex(bob.GetPublicKey(pk), 32) == "de9edb7d7b7dc1b4d35b61c2ece435373f8343c85b78674dadfc7e146f882b4f", "bob public key");
    alice.CreateShare(bob.GetPublicKey(0), sh1);
    bob.CreateShare(alice.GetPublicKey(0), sh2);
    check(hex(sh1, 32) == "4a5d9d5ba4ce2de1728e3bf480350f25e07e21c947d19e3376f09b3c1e161742" && !memcmp(sh1, sh2, 32), "shared secret, both sides");
    alice.CreateSharedKey(bob.GetPublicKey(0), k1, 64);
    bob.CreateSharedKey(alice.GetPublicKey(0), k2, 48);
    unsigned char dg[SHA512_DIGEST_LENGTH];
    SHA512_CTX h;
    SHA512_Init(&h);
    SHA512_Update(&h, sh1, 32);
    SHA512_Final(dg, &h);
    check(!memcmp(k1, dg, 64) && !memcmp(k2, dg, 48), "CreateSharedKey = SHA-512(DH secret), 64- and 48-byte forms");

    // ---- ED25519Private / ED25519Public: RFC 8032 7.1 ----
    struct { const char *sk, *pk, *msg, *sig; } v[3] = {
        { "9d61b19deffd5a60ba844af492ec2cc44449c5697b326919703bac031cae7f60",
          "d75a980182b10ab7d54bfed3c964073a0ee172f3daa62325af021a68f707511a", "",
          "e5564300c360ac729086e2cc806e828a84877f1eb8e5d974d873e065224901555fb8821590a33bacc61e39701cf9b46bd25bf5f0595bbe24655141438e7a100b" },
        { "4ccd089b28ff96da9db6c346ec114e0f5b8a319f35aba624da8cf6ed4fb8a6fb",
          "3d4017c3e843895a92b70aa74d1b7ebc9c982ccf2ec4968cc0cd55f12af4660c", "72",
          "92a009a9f0d4cab8720e820b5f642540a2b27b5416503f8fb3762223ebdb69da085ac1e43e15996e458f3613d0f11d8c387b2eaeb4302aeeb00d291612bb0c00" },
        { "c5aa8df43f9f837bedb7442f31dcb7b166d38535076f094b85ce3a2e0b4458f7",
          "fc51cd8e6218a1a38da47ed00230f0580816ed13ba3303ac5deb911548908025", "af82",
          "6291d657deec24024827e69c3abe01a30ce548a284743a445e3680d7db5ac3ac18ff9b538d16f290ae67f760984dc6594a7c15e9716ed28dc027beceea1ec40a" },
    };
    for (int i = 0; i < 3; i++) {
        unsigned char sk[32], msg[8], sig[64];
        unhex(sk, v[i].sk);
        const unsigned n = (unsigned)strlen(v[i].msg) / 2;
        unhex(msg, v[i].msg);
        ED25519Private priv(sk, 32);                      // CreateKeyPair with edp_genkey_blinding
        check(hex(priv.GetPublicKey(), 32) == v[i].pk, "RFC 8032 public key (blinded keygen, reference's static context)");
        priv.SignMessage(msg, n, sig);                    // SignMessage with edp_signature_blinding
        check(hex(sig, 64) == v[i].sig, "RFC 8032 signature (blinded sign, reference's static context)");
        ED25519Private reload(priv.GetPrivateKey(), 64);
        unsigned char sig2[64];
        reload.SignMessage(msg, n, sig2);
        check(!memcmp(sig, sig2, 64), "private key reloaded from its 64-byte form signs identically");
        ED25519Public pub(priv.GetPublicKey());
        check(pub.VeifySignature(msg, n, sig), "signature verifies");
        sig[7] ^= 4;
        check(!pub.VeifySignature(msg, n, sig), "corrupted signature is rejected");
    }
    ED25519Private rnd(0, 0);                              // random key path (GetRandomBytes)
    unsigned char m[5] = { 1, 2, 3, 4, 5 }, s[64];
    rnd.SignMessage(m, 5, s);
    check(ED25519Public(rnd.GetPublicKey()).VeifySignature(m, 5, s), "random key: sign -> verify");
    printf("%d failure(s)\n", failures);
    return failures;
}
