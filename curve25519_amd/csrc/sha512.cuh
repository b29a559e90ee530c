// curve25519_amd/csrc/sha512.cuh -- SHA-512 per lane, for the hashes inside Ed25519 keygen / sign /
// verify (the role of source/sha512.c: SHA512_Init :50, SHA512_Update :118, SHA512_Final :67 as used at
// ed25519_sign.c:355-357, :385-395, :404-408 and ed25519_verify.c:298-302).
//
// Every hash on this path is  H(prefix || message)  where the prefix is 32 or 64 bytes that already sit
// in registers (a secret seed, enc(R) || pk, ...) and the message is `len` bytes in global memory.  The
// prefix is whole 64-bit words, so the first block is assembled statically and the message is streamed
// word by word with the FIPS 180-4 padding generated on the fly.  `len` is uniform across the batch, so
// the block loop does not diverge.
#pragma once
#include "fe25519.cuh"

namespace c25519 {

__device__ constexpr u64 SHA512_K[80] = {
    0x428a2f98d728ae22ull, 0x7137449123ef65cdull, 0xb5c0fbcfec4d3b2full, 0xe9b5dba58189dbbcull,
    0x3956c25bf348b538ull, 0x59f111f1b605d019ull, 0x923f82a4af194f9bull, 0xab1c5ed5da6d8118ull,
    0xd807aa98a3030242ull, 0x12835b0145706fbeull, 0x243185be4ee4b28cull, 0x550c7dc3d5ffb4e2ull,
    0x72be5d74f27b896full, 0x80deb1fe3b1696b1ull, 0x9bdc06a725c71235ull, 0xc19bf174cf692694ull,
    0xe49b69c19ef14ad2ull, 0xefbe4786384f25e3ull, 0x0fc19dc68b8cd5b5ull, 0x240ca1cc77ac9c65ull,
    0x2de92c6f592b0275ull, 0x4a7484aa6ea6e483ull, 0x5cb0a9dcbd41fbd4ull, 0x76f988da831153b5ull,
    0x983e5152ee66dfabull, 0xa831c66d2db43210ull, 0xb00327c898fb213full, 0xbf597fc7beef0ee4ull,
    0xc6e00bf33da88fc2ull, 0xd5a79147930aa725ull, 0x06ca6351e003826full, 0x142929670a0e6e70ull,
    0x27b70a8546d22ffcull, 0x2e1b21385c26c926ull, 0x4d2c6dfc5ac42aedull, 0x53380d139d95b3dfull,
    0x650a73548baf63deull, 0x766a0abb3c77b2a8ull, 0x81c2c92e47edaee6ull, 0x92722c851482353bull,
    0xa2bfe8a14cf10364ull, 0xa81a664bbc423001ull, 0xc24b8b70d0f89791ull, 0xc76c51a30654be30ull,
    0xd192e819d6ef5218ull, 0xd69906245565a910ull, 0xf40e35855771202aull, 0x106aa07032bbd1b8ull,
    0x19a4c116b8d2d0c8ull, 0x1e376c085141ab53ull, 0x2748774cdf8eeb99ull, 0x34b0bcb5e19b48a8ull,
    0x391c0cb3c5c95a63ull, 0x4ed8aa4ae3418acbull, 0x5b9cca4f7763e373ull, 0x682e6ff3d6b2b8a3ull,
    0x748f82ee5defb2fcull, 0x78a5636f43172f60ull, 0x84c87814a1f0ab72ull, 0x8cc702081a6439ecull,
    0x90befffa23631e28ull, 0xa4506cebde82bde9ull, 0xbef9a3f7b2c67915ull, 0xc67178f2e372532bull,
    0xca273eceea26619cull, 0xd186b8c721c0c207ull, 0xeada7dd6cde0eb1eull, 0xf57d4f7fee6ed178ull,
    0x06f067aa72176fbaull, 0x0a637dc5a2c898a6ull, 0x113f9804bef90daeull, 0x1b710b35131c471bull,
    0x28db77f523047d84ull, 0x32caab7b40c72493ull, 0x3c9ebe0a15c9bebcull, 0x431d67c49c100d4cull,
    0x4cc5d4becb3e42b6ull, 0x597f299cfc657e2aull, 0x5fcb6fab3ad6faecull, 0x6c44198c4a475817ull,
};

// 64-bit rotate right by a compile-time n (1..63) as two v_alignbit_b32 over the halves: the compiler expands the plain
// shift-or form into a 64-bit right shift, a left shift and one or two ORs (four instructions per rotate, ten rotates
// per round)
C25519_DEV u64 rotr64(u64 x, int n)
{
    const u32 lo = (u32)x, hi = (u32)(x >> 32);
    if (n == 32) return ((u64)lo << 32) | hi;
    const u32 a = n < 32 ? lo : hi, b = n < 32 ? hi : lo;    // rotate (b:a) right by n mod 32
    const int r = n & 31;
    return ((u64)alignbit32(a, b, r) << 32) | alignbit32(b, a, r);
}

// the round's three-input functions on 64-bit words: one v_bitop3_b32 per half (valu_gfx950.cuh).  pair64 keeps the two halves
// ONE 64-bit value for the additions that follow (left as (hi << 32) | lo, the compiler adds the halves as two 64-bit terms)
C25519_DEV u64 xor3_64(u64 a, u64 b, u64 c)
{
    return pair64(xor3_32((u32)a, (u32)b, (u32)c), xor3_32((u32)(a >> 32), (u32)(b >> 32), (u32)(c >> 32)));
}
C25519_DEV u64 ch_64(u64 e, u64 f, u64 g)
{
    return pair64(ch_32((u32)e, (u32)f, (u32)g), ch_32((u32)(e >> 32), (u32)(f >> 32), (u32)(g >> 32)));
}
C25519_DEV u64 maj_64(u64 a, u64 b, u64 c)
{
    return pair64(maj_32((u32)a, (u32)b, (u32)c), maj_32((u32)(a >> 32), (u32)(b >> 32), (u32)(c >> 32)));
}

// big-endian 64-bit word from two little-endian 32-bit words as they sit in memory
C25519_DEV u64 be64_from_le32(u32 lo_addr_word, u32 hi_addr_word)
{
    return ((u64)__builtin_bswap32(lo_addr_word) << 32) | __builtin_bswap32(hi_addr_word);
}

template <bool SCHEDULE>
C25519_DEV void sha512_rounds16(u64 (&v)[8], u64 (&w)[16], int r)
{
    u64 a = v[0], b = v[1], c = v[2], d = v[3], e = v[4], f = v[5], g = v[6], h = v[7];
#pragma unroll
    for (int i = 0; i < 16; i++) {
        if (SCHEDULE) {
            const u64 w15 = w[(i + 1) & 15], w2 = w[(i + 14) & 15];
            const u64 s0 = xor3_64(rotr64(w15, 1), rotr64(w15, 8), w15 >> 7);
            const u64 s1 = xor3_64(rotr64(w2, 19), rotr64(w2, 61), w2 >> 6);
            w[i] += s0 + w[(i + 9) & 15] + s1;
        }
        const u64 S1 = xor3_64(rotr64(e, 14), rotr64(e, 18), rotr64(e, 41));
        const u64 ch = ch_64(e, f, g);
        const u64 t1 = h + S1 + ch + SHA512_K[r + i] + w[i];
        const u64 S0 = xor3_64(rotr64(a, 28), rotr64(a, 34), rotr64(a, 39));
        const u64 mj = maj_64(a, b, c);
        const u64 t2 = S0 + mj;
        h = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    v[0] = a; v[1] = b; v[2] = c; v[3] = d; v[4] = e; v[5] = f; v[6] = g; v[7] = h;
}

C25519_DEV void sha512_compress(u64 (&st)[8], u64 (&w)[16])
{
    u64 v[8];
#pragma unroll
    for (int i = 0; i < 8; i++) v[i] = st[i];
    sha512_rounds16<false>(v, w, 0);
#pragma unroll 1
    for (int r = 16; r < 80; r += 16) sha512_rounds16<true>(v, w, r);
#pragma unroll
    for (int i = 0; i < 8; i++) st[i] += v[i];
}

// sixteen rounds whose schedule words come ready-made, K already added (wk[i] = W[r + i] + K[r + i]): 28 instructions a round
// instead of 47 -- the rounds' share of a compression whose schedule another wave computes (coop25519.cuh: ShaTwoWaves)
C25519_DEV void sha512_rounds16_wk(u64 (&v)[8], const u64* wk)
{
    u64 a = v[0], b = v[1], c = v[2], d = v[3], e = v[4], f = v[5], g = v[6], h = v[7];
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const u64 S1 = xor3_64(rotr64(e, 14), rotr64(e, 18), rotr64(e, 41));
        const u64 t1 = h + S1 + ch_64(e, f, g) + wk[i];
        const u64 S0 = xor3_64(rotr64(a, 28), rotr64(a, 34), rotr64(a, 39));
        const u64 t2 = S0 + maj_64(a, b, c);
        h = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    v[0] = a; v[1] = b; v[2] = c; v[3] = d; v[4] = e; v[5] = f; v[6] = g; v[7] = h;
}
// the next sixteen schedule words from the last sixteen (in place), each stored as W + K for sha512_rounds16_wk
C25519_DEV void sha512_schedule16_wk(u64 (&w)[16], u64* wk_out, int r)
{
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const u64 w15 = w[(i + 1) & 15], w2 = w[(i + 14) & 15];
        const u64 s0 = xor3_64(rotr64(w15, 1), rotr64(w15, 8), w15 >> 7);
        const u64 s1 = xor3_64(rotr64(w2, 19), rotr64(w2, 61), w2 >> 6);
        w[i] += s0 + w[(i + 9) & 15] + s1;
        wk_out[i] = w[i] + SHA512_K[r + i];
    }
}

// how a block is compressed: by the calling lane alone (every kernel but the per-wave fixed-base ones)
struct ShaPlain {
    C25519_DEV void compress(u64 (&st)[8], u64 (&w)[16]) const { sha512_compress(st, w); }
};
// blocks SHA-512(prefix of PW words || len message bytes) takes: what a helper that serves the compressions must count
C25519_DEV int sha512_blocks(int prefix_words, size_t len) { return (int)((8 * (size_t)prefix_words + len + 17 + 127) / 128); }

// m-th 64-bit big-endian word of  message || 0x80 || 0...  (without the trailing length words)
C25519_DEV u64 sha512_msg_word(const uint8_t* msg, size_t len, size_t m)
{
    const size_t o = 8 * m;
    u64 v = 0;
    if (o + 8 <= len) {
        if ((reinterpret_cast<uintptr_t>(msg + o) & 3u) == 0) {
            const u32* p = reinterpret_cast<const u32*>(msg + o);
            return be64_from_le32(p[0], p[1]);
        }
#pragma unroll
        for (int j = 0; j < 8; j++) v = (v << 8) | msg[o + j];
        return v;
    }
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const size_t q = o + j;
        const u64 byte = q < len ? msg[q] : (q == len ? 0x80u : 0u);
        v = (v << 8) | byte;
    }
    return v;
}

// digest = SHA-512(prefix[0..PW) as big-endian words || msg[0..len)).  PW = 4 or 8.
template <int PW, typename Sha = ShaPlain>
C25519_DEV void sha512_prefixed(u64 (&digest)[8], const u64 (&prefix)[PW], const uint8_t* msg, size_t len, const Sha& sha = Sha())
{
    u64 st[8] = { 0x6a09e667f3bcc908ull, 0xbb67ae8584caa73bull, 0x3c6ef372fe94f82bull, 0xa54ff53a5f1d36f1ull,
                  0x510e527fade682d1ull, 0x9b05688c2b3e6c1full, 0x1f83d9abfb41bd6bull, 0x5be0cd19137e2179ull };
    const size_t total = 8 * PW + len;
    const size_t nblocks = (total + 17 + 127) / 128;
    const size_t last_word = 16 * nblocks - 1;          // holds the bit length (low 64 bits)
    u64 w[16];
#pragma unroll 1
    for (size_t blk = 0; blk < nblocks; blk++) {
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const size_t g = 16 * blk + j;               // word index in the padded stream
            u64 v;
            if (j < PW && blk == 0) v = prefix[j];
            else if (g == last_word) v = (u64)total << 3;
            else if (g == last_word - 1) v = (u64)total >> 61;
            else v = sha512_msg_word(msg, len, g - PW);
            w[j] = v;
        }
#ifdef C25519_SHA_LOW_PRIO                               // A/B knob: the compression (no multiplies) as a low-priority run
        C25519_VOP2_RUN_BEGIN();
#endif
        sha.compress(st, w);
#ifdef C25519_SHA_LOW_PRIO
        C25519_VOP2_RUN_END();
#endif
    }
#pragma unroll
    for (int i = 0; i < 8; i++) digest[i] = st[i];
}

// digest words (big-endian u64) -> 16 little-endian u32 words of the 64-byte digest string
C25519_DEV void sha512_digest_le_words(u32 (&out)[16], const u64 (&digest)[8])
{
#pragma unroll
    for (int i = 0; i < 8; i++) {
        out[2 * i] = __builtin_bswap32((u32)(digest[i] >> 32));
        out[2 * i + 1] = __builtin_bswap32((u32)digest[i]);
    }
}

// 8 little-endian u32 words (32 bytes as they sit in memory) -> 4 big-endian u64 stream words
C25519_DEV void sha512_words_from_le32(u64* dst, const u32 (&w)[8])
{
#pragma unroll
    for (int i = 0; i < 4; i++) dst[i] = be64_from_le32(w[2 * i], w[2 * i + 1]);
}

}  // namespace c25519
