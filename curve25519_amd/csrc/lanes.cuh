// curve25519_amd/csrc/lanes.cuh -- what ONE lane does for each operation of the path, between its loads and its
// stores: record I/O, the hashing / scalar steps around the scalar multiplications, and the unit-test operations of
// the self-test hooks.  The kernels of engine.hip are thin wrappers (indexing, LDS staging, scratch) around these;
// the CPU unit tests (tests/host_emul/) drive the same functions one lane at a time.
//
// Reference counterparts: ed25519_CreateKeyPair (source/ed25519_sign.c:344-367), ed25519_SignMessage (:372-419),
// ed25519_Verify_Check (source/ed25519_verify.c:287-313), edp_BasePointMultiply with a blinding context
// (ed25519_sign.c:246-268) and ed25519_Blinding_Init (:289-331).
#pragma once
#include "fe25519.cuh"
#include "ge25519.cuh"
#include "sc25519.cuh"
#include "sha512.cuh"
#include "x25519.cuh"

namespace c25519 {

// ---- record I/O ---------------------------------------------------------------------------------------------
// 32-byte API records as two 16-byte accesses (a wave covers 2 KiB of contiguous memory)
C25519_DEV void load32(u32 (&w)[8], const void* base, size_t i)
{
    const uint4* p = reinterpret_cast<const uint4*>(base) + 2 * i;
    const uint4 a = p[0], b = p[1];
    w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w; w[4] = b.x; w[5] = b.y; w[6] = b.z; w[7] = b.w;
}
C25519_DEV void store32(void* base, size_t i, const u32 (&w)[8])
{
    uint4* p = reinterpret_cast<uint4*>(base) + 2 * i;
    p[0] = make_uint4(w[0], w[1], w[2], w[3]);
    p[1] = make_uint4(w[4], w[5], w[6], w[7]);
}
// scratch arrays are struct-of-arrays: word w of element i at base[w*n + i], so every access by a wave
// is one contiguous 256-byte segment whichever element -> lane mapping a kernel uses
C25519_DEV void soa_store_fe(u32* base, size_t n, size_t i, const fe& f)
{
#pragma unroll
    for (int w = 0; w < 10; w++) base[(size_t)w * n + i] = f.v[w];
}
C25519_DEV void soa_load_fe(fe& f, const u32* base, size_t n, size_t i)
{
#pragma unroll
    for (int w = 0; w < 10; w++) f.v[w] = base[(size_t)w * n + i];
}
C25519_DEV void soa_store8(u32* base, size_t n, size_t i, const u32 (&v)[8])
{
#pragma unroll
    for (int w = 0; w < 8; w++) base[(size_t)w * n + i] = v[w];
}
C25519_DEV void soa_load8(u32 (&v)[8], const u32* base, size_t n, size_t i)
{
#pragma unroll
    for (int w = 0; w < 8; w++) v[w] = base[(size_t)w * n + i];
}

// The input records of a call of ONE element, carried in the kernel's arguments (engine.hip: call_words): the first loads of a
// zero-copy call read pinned host memory over PCIe -- ~1.1 us before anything can start (profiles/r06_launch_latency.txt) --,
// the arguments arrive with the dispatch.  use == 0: the records are read from memory as in every other call.
// X25519: pk in words 0..7, sk in 8..15 (coop_ops.cuh reads them straight from the arguments).  The fixed-base operations: the
// record (sk, or priv) from word 0, the message from word 16; the kernel lays them down in LDS and the per-wave code reads them
// there through the pointers it is given (stage_call_words).
constexpr int CALL_WORDS = 32;
struct CallWords { u32 w[CALL_WORDS]; u32 use; };

// messages of a batch: fixed stride (offsets == nullptr) or ragged (message i = base[offsets[i] .. offsets[i+1]))
struct Msgs {
    const uint8_t* base;
    size_t fixed;
    const unsigned long long* offsets;
    C25519_DEV const uint8_t* ptr(size_t i) const { return base + (offsets ? (size_t)offsets[i] : i * fixed); }
    C25519_DEV size_t len(size_t i) const { return offsets ? (size_t)(offsets[i + 1] - offsets[i]) : fixed; }
};

// ---- Ed25519 hashing / scalar steps ---------------------------------------------------------------------------
// a = clamp(first half of SHA-512(seed)); the second half as 4 big-endian stream words   (ed25519_sign.c:355-360)
template <typename Sha = ShaPlain>
C25519_DEV void ed_expand_seed(u32 (&a)[8], u64 (&b_words)[4], const u32 (&seed)[8], const Sha& sha = Sha())
{
    u64 pre[4], dg[8];
    sha512_words_from_le32(pre, seed);
    sha512_prefixed<4>(dg, pre, nullptr, 0, sha);
    u32 le[16];
    sha512_digest_le_words(le, dg);
#pragma unroll
    for (int i = 0; i < 8; i++) a[i] = le[i];
    clamp_words(a);
#pragma unroll
    for (int i = 0; i < 4; i++) b_words[i] = dg[4 + i];
}

// first part of ed25519_SignMessage (:385-397): a = clamp(H(sk)[0..31]), r = H(H(sk)[32..63] || m) mod L, canonical
template <typename Sha = ShaPlain>
C25519_DEV void ed_sign_nonce(u32 (&a)[8], u32 (&r)[8], const u32 (&seed)[8], const uint8_t* msg, size_t len, const Sha& sha = Sha())
{
    u64 b_words[4], dg[8];
    u32 le[16];
    ed_expand_seed(a, b_words, seed, sha);
    sha512_prefixed<4>(dg, b_words, msg, len, sha);
    sha512_digest_le_words(le, dg);
    sc_reduce512(r, le);
    sc_mod(r);
}

// h = H(enc(R) || pk || m) reduced to 256 bits, congruent mod L (not canonical)   (:404-409 / ed25519_verify.c:298-305)
template <typename Sha = ShaPlain>
C25519_DEV void ed_hram(u32 (&h)[8], const u32 (&encR)[8], const u32 (&pkw)[8], const uint8_t* msg, size_t len, const Sha& sha = Sha())
{
    u32 le[16];
    u64 pre[8], dg[8];
    sha512_words_from_le32(pre, encR);
    sha512_words_from_le32(pre + 4, pkw);
    sha512_prefixed<8>(dg, pre, msg, len, sha);
    sha512_digest_le_words(le, dg);
    sc_reduce512(h, le);
}

// last part of ed25519_SignMessage (:404-414): S = H(enc(R) || pk || m) * a + r mod L, canonical
template <typename Sha = ShaPlain>
C25519_DEV void ed_sign_s(u32 (&s)[8], const u32 (&encR)[8], const u32 (&pkw)[8], const uint8_t* msg, size_t len,
                          const u32 (&a)[8], const u32 (&r)[8], const Sha& sha = Sha())
{
    u32 h[8];
    ed_hram(h, encR, pkw, msg, len, sha);
    sc_mul(s, h, a);
    sc_add(s, s, r);
    sc_mod(s);
}

// -A from the 32 key bytes: y as given (bit 255 stripped), x with the INVERTED parity, no validation
// (ed25519_Verify_Init :191-197)
C25519_DEV void ed_decode_neg_key(ge_ext& Q, const u32 (&pkw)[8])
{
    u32 yw[8];
#pragma unroll
    for (int i = 0; i < 8; i++) yw[i] = pkw[i];
    const u32 parity = yw[7] >> 31;
    yw[7] &= 0x7fffffffu;
    fe_from_words(Q.Y, yw);
    ge_calc_x(Q.X, Q.Y, ~parity);
    fe_mul(Q.T, Q.X, Q.Y);
    fe_set_u32(Q.Z, 1);
}

// ---- blinding (ed25519_sign.c:254-259, :289-331) ------------------------------------------------------------
// A blinding context is the reference's EDP_BLINDING_CTX shape, 192 bytes as 48 little-endian words:
//   bl[8]    scalar, L - t                          (added to the secret scalar before the walk)
//   zr[8]    256 random bits                        (projective Z of the walk's starting point)
//   BP[32]   t*B as a PE_POINT, four canonical elements YpX, YmX, T2d, Z2   (added after the walk)
// so that (k + bl)*B + BP = k*B: the table lookups and the walk see a scalar that changes with every context.
constexpr int BLIND_WORDS = 48;

// S = k*B computed as (k + bl)*B + BP with the starting point's Z randomised   (edp_BasePointMultiply, blinding != 0).
// ctx: the 48 context words (wave-uniform).  Only bl and zr are read before the walk; the four fields of BP are fetched
// one at a time for the final addition (held across the walk they cost 40 registers the 1024-thread kernels do not have).
// k + bl is reduced mod L before the recoding: sc_signed_comb adds L to an even scalar in 256 bits, so it needs its input
// below 2^256 - L; contexts from ed25519_Blinding_Init have bl <= L, but a context is caller-supplied bytes.
// base_mult(S, t, zr): S = t * B from the starting point spread by zr, T of the result included (either comb).
template <typename BaseMult>
C25519_DEV void ge_base_mult_blinded_with(ge_ext& S, const u32 (&k)[8], const u32* ctx, BaseMult base_mult)
{
    u32 t[8], w[8];
    {
        u32 bl[8];
#pragma unroll
        for (int i = 0; i < 8; i++) bl[i] = ctx[i];
        sc_add(t, k, bl);                     // 256 bits, congruent to k + bl mod L (eco_AddReduce :255)
        sc_mod(t);
    }
    {
        fe zr;
#pragma unroll
        for (int i = 0; i < 8; i++) w[i] = ctx[8 + i];
        fe_from_words(zr, w);
        base_mult(S, t, zr);                  // T of the result feeds the addition below
    }
    // S += BP (:257); the affine conversion that follows never reads T.  ge_add_pe with the fields streamed in.
    fe q, a, b, e, f, g, h;
    auto field = [&](int j) {
#pragma unroll
        for (int i = 0; i < 8; i++) w[i] = ctx[16 + 8 * j + i];
        fe_from_words(q, w);
    };
    fe_sub(a, S.Y, S.X);
    field(1);  fe_mul(a, a, q);               // (Y - X) * YmX
    fe_add(b, S.Y, S.X);
    field(0);  fe_mul(b, b, q);               // (Y + X) * YpX
    fe_sub(e, b, a);
    fe_add(h, b, a);
    field(2);  fe_mul(a, S.T, q);             // C = T * T2d
    field(3);  fe_mul(b, S.Z, q);             // D = Z * Z2
    fe_sub(f, b, a);
    fe_add(g, b, a);
    fe_mul(S.X, e, f);
    fe_mul(S.Z, g, f);
    fe_mul(S.Y, g, h);
}

C25519_DEV void ge_base_mult_blinded(ge_ext& S, const u32 (&k)[8], const u32* ctx, const u32* lds_tbl)
{
    ge_base_mult_blinded_with(S, k, ctx, [&](ge_ext& P, const u32 (&t)[8], const fe& zr) { ge_base_mult<true>(P, t, lds_tbl, &zr); });
}

// ---- 8-fold base table rows ----------------------------------------------------------------------------------
// row k = 2^extra * (sum over set bits i of k of 2^(32 i) * B) as canonical words of (Y+X, Y-X, 2dT): with extra = 0
// the content of the reference's source/base_folding8.h, derived from B by doubling/adding (the recipe of
// test/curve25519_selftest.c:498-551); extra > 0 gives the shifted tables of ge_base_mult's short walk.
C25519_DEV void ge_base_table_row(u32 (&rows)[3][8], u32 k, int extra)
{
    ge_pa B;
    B.ypx = fe_const(K_BY); B.ymx = fe_const(K_BY);
    {
        fe t;
        fe_add(t, B.ypx, fe_const(K_BX)); fe_carry32(B.ypx, t);
        fe_sub(t, B.ymx, fe_const(K_BX)); fe_carry32(B.ymx, t);
    }
    B.t2d = fe_const(K_BT2D);

    ge_ext S;                                 // neutral element (0 : 1 : 1 : 0)
    fe_set_u32(S.X, 0); fe_set_u32(S.Y, 1); fe_set_u32(S.Z, 1); fe_set_u32(S.T, 0);
#pragma unroll 1
    for (int i = 7; i >= 0; i--) {            // Horner over the 8 index bits, 32 doublings apart
        if ((k >> i) & 1) ge_add_pa(S, B);
        const int dbl = i ? 32 : extra;
#pragma unroll 1
        for (int j = 0; j < dbl; j++) ge_double(S);
    }
    fe zi, x, y, t, row[3];
    fe_invert(zi, S.Z);
    fe_mul(x, S.X, zi);
    fe_mul(y, S.Y, zi);
    fe_add(row[0], y, x);
    fe_sub(row[1], y, x);
    fe_mul(t, x, y);
    fe_mul(row[2], t, fe_const(K_2D));
#pragma unroll
    for (int f = 0; f < 3; f++) fe_to_words(rows[f], row[f]);
}

// row idx (teeth - 1 bits) of a signed comb table with `teeth` teeth `spacing` bits apart:
// 2^extra * (2^(spacing*(teeth-1)) + sum over j < teeth-1 of (bit j of idx ? + : -) 2^(spacing j)) * B, as canonical words
// of (Y+X, Y-X, 2dT) -- what the signed recodings (ge_base_mult: 8 x 32; the verification walk: SC_TEETH x SC_COLS) select
// for a column whose top digit is +1
// ... of ANY point P given in affine precomputed form (ge_signed_comb_row below: the base point; the two-phase
// verification's per-key comb: -A)
C25519_DEV void ge_signed_comb_row_of(u32 (&rows)[3][8], const ge_pa& B, u32 idx, int extra, int teeth, int spacing)
{
    ge_pa Bn;
    Bn.ypx = B.ymx; Bn.ymx = B.ypx;                          // -P
    { fe t; fe_neg(t, B.t2d); fe_carry32(Bn.t2d, t); }

    ge_ext S;                                                // top tooth: + P
    fe_set_u32(S.X, 0); fe_set_u32(S.Y, 1); fe_set_u32(S.Z, 1); fe_set_u32(S.T, 0);
    ge_add_pa(S, B);
#pragma unroll 1
    for (int i = teeth - 2; i >= 0; i--) {                   // Horner over the signed teeth, `spacing` doublings apart
#pragma unroll 1
        for (int j = 0; j < spacing; j++) ge_double(S);
        if ((idx >> i) & 1) ge_add_pa(S, B);
        else ge_add_pa(S, Bn);
    }
#pragma unroll 1
    for (int j = 0; j < extra; j++) ge_double(S);
    fe zi, x, y, t, row[3];
    fe_invert(zi, S.Z);
    fe_mul(x, S.X, zi);
    fe_mul(y, S.Y, zi);
    fe_add(row[0], y, x);
    fe_sub(row[1], y, x);
    fe_mul(t, x, y);
    fe_mul(row[2], t, fe_const(K_2D));
#pragma unroll
    for (int f = 0; f < 3; f++) fe_to_words(rows[f], row[f]);
}

C25519_DEV void ge_signed_comb_row(u32 (&rows)[3][8], u32 idx, int extra, int teeth = 8, int spacing = 32)
{
    ge_pa B;
    B.ypx = fe_const(K_BY); B.ymx = fe_const(K_BY);
    {
        fe t;
        fe_add(t, B.ypx, fe_const(K_BX)); fe_carry32(B.ypx, t);
        fe_sub(t, B.ymx, fe_const(K_BX)); fe_carry32(B.ymx, t);
    }
    B.t2d = fe_const(K_BT2D);
    ge_signed_comb_row_of(rows, B, idx, extra, teeth, spacing);
}

// ed25519_Blinding_Init (ed25519_sign.c:289-331) for one context: digest = SHA-512(domain || seed),
// t = digest[0..31] mod L, bl = L - t, zr = digest[32..63], BP = PE(t*B) (affine, Z2 = 2).  The 32-byte domain string
// "c25519_amd_blinding_cxv1--------" takes the place of the reference's compiled-in custom blinder (custom_blind.c),
// which likewise only seeds the derivation.  lds_tbl: the BASE_NT staged signed comb tables.
// the context's scalars: t = digest[0..31] mod L (canonical; BP = t * B), bl = L - t, zr = digest[32..63]
C25519_DEV void ed_blinding_scalars(u32 (&t)[8], u32 (&bl)[8], u32 (&zr)[8], const uint8_t* seed, size_t seed_len)
{
    const u32 domain[8] = { 0x35353263u, 0x615f3931u, 0x625f646du, 0x646e696cu, 0x5f676e69u, 0x31767863u, 0x2d2d2d2du, 0x2d2d2d2du };
    u64 pre[4], dg[8];
    u32 le[16];
    sha512_words_from_le32(pre, domain);
    sha512_prefixed<4>(dg, pre, seed, seed_len);
    sha512_digest_le_words(le, dg);
#pragma unroll
    for (int i = 0; i < 8; i++) { t[i] = le[i]; zr[i] = le[8 + i]; }
    sc_mod(t);                                            // eco_Mod (:316)
    u32 borrow = 0;                                       // bl = L - t (:317)
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const u64 d = (u64)K_L[i] - t[i] - borrow;
        bl[i] = (u32)d;
        borrow = (u32)(d >> 63);
    }
}

// the 48 context words from the scalars and the canonical affine coordinates of BP = t * B: BP as (y+x, y-x, 2dxy, 2)
C25519_DEV void ed_blinding_store(u32* ctx, const u32 (&bl)[8], const u32 (&zr)[8], const u32 (&xw)[8], const u32 (&yw)[8])
{
    u32 w[8];
    fe x, y, ypx, ymx, t2d, tmp;
    fe_from_words(x, xw);
    fe_from_words(y, yw);
    fe_add(ypx, y, x);
    fe_sub(ymx, y, x);
    fe_mul(tmp, x, y);
    fe_mul(t2d, tmp, fe_const(K_2D));
#pragma unroll
    for (int i = 0; i < 8; i++) { ctx[i] = bl[i]; ctx[8 + i] = zr[i]; }
    fe_to_words(w, ypx);
#pragma unroll
    for (int i = 0; i < 8; i++) ctx[16 + i] = w[i];
    fe_to_words(w, ymx);
#pragma unroll
    for (int i = 0; i < 8; i++) ctx[24 + i] = w[i];
    fe_to_words(w, t2d);
#pragma unroll
    for (int i = 0; i < 8; i++) ctx[32 + i] = w[i];
#pragma unroll
    for (int i = 0; i < 8; i++) ctx[40 + i] = i == 0 ? 2u : 0u;      // Z2 = 2 (affine BP)
}

C25519_DEV void ed_blinding_init_lane(u32* ctx, const uint8_t* seed, size_t seed_len, const u32* lds_tbl)
{
    u32 t[8], bl[8], zr[8], xw[8], yw[8];
    ed_blinding_scalars(t, bl, zr, seed, seed_len);
    ge_ext S;
    ge_base_mult(S, t, lds_tbl);                          // T = t*B (:319-321 without the bootstrap blinder)
    ge_to_affine_words(xw, yw, S);
    ed_blinding_store(ctx, bl, zr, xw, yw);
}

// ---- unit-test operations ---------------------------------------------------------------------------------------
// field hook: r = op(x, y), see include/curve25519_amd.h c25519_amd_fe_selftest
C25519_DEV void fe_selftest_op(u32 (&ow)[8], const u32 (&aw)[8], const u32 (&bw)[8], int op)
{
    fe x, y, r, t;
    fe_from_words(x, aw);
    fe_from_words(y, bw);
    switch (op) {
    case 0: fe_mul(r, x, y); break;
    case 1: fe_sqr(r, x); break;
    case 2: fe_add(r, x, y); break;
    case 3: fe_sub(r, x, y); break;
    case 4: fe_invert(r, x); break;
    case 5: fe_pow2523(r, x); break;
    case 6: r = x; break;
    case 7: fe_sub(r, x, y); fe_add(t, x, y); fe_mul(r, r, t); break;
    case 8: fe_sqr_sub(r, x, y); break;                       // x^2 - y
    case 9: fe_add(t, x, y); fe_sqr2_add_sub(r, x, t, y); break;   // 2x^2 + (x+y) - y
    case 10: fe_mul121665_add(r, x, y); break;                // x + 121665 y
    case 12: fe_invert_fermat(r, x); break;                   // 1/x the reference's way: x^(p-2)
    case 13: fe_invert_safegcd(r, x); break;                  // 1/x by division steps (safegcd25519.cuh)
    case 14: fe_invert_quad(r, x); break;                     // ... with a quad of lanes on the same x (k_fe_selftest_quad)
    default: fe_mul_small(r, x, 9); break;                    // 9 x
    }
    fe_to_words(ow, r);
}

// scalar hook (mod L): the device side of the reference's eco_* unit checks (test/curve25519_selftest.c:624-714).
// a = 16 words (512 bits), b = 8 words; "raw" results are 256 bits congruent to the exact value mod L.
//   0 canonical(a mod L)            sc_reduce512 + sc_mod          (eco_DigestToWords + eco_Mod)
//   1 raw  a mod L                  sc_reduce512
//   2 canonical(a[0..7] mod L)      sc_mod                         (eco_Mod; needs a[0..7] < 2^256 only)
//   3 raw  a[0..7] * b              sc_mul                         (eco_MulReduce)
//   4 raw  a[0..7] + b              sc_add                         (eco_AddReduce)
//   5 raw  a[8] * 2^256 + a[0..7]   sc_reduce_hi                   (eco_ReduceHiWord)
//   6 canonical(a[0..7] * b + a[8..15])   the S = h*a + r step of signing
C25519_DEV void sc_selftest_op(u32 (&ow)[8], const u32 (&aw)[16], const u32 (&bw)[8], int op)
{
    u32 t[16], x[8], y[8];
#pragma unroll
    for (int i = 0; i < 16; i++) t[i] = aw[i];
#pragma unroll
    for (int i = 0; i < 8; i++) { x[i] = aw[i]; y[i] = aw[8 + i]; }
    switch (op) {
    case 0: sc_reduce512(ow, t); sc_mod(ow); break;
    case 1: sc_reduce512(ow, t); break;
    case 2: sc_mod(x);
#pragma unroll
        for (int i = 0; i < 8; i++) ow[i] = x[i];
        break;
    case 3: sc_mul(ow, x, bw); break;
    case 4: sc_add(ow, x, bw); break;
    case 5: sc_reduce_hi(ow, aw[8], x); break;
    default: sc_mul(ow, x, bw); sc_add(ow, ow, y); sc_mod(ow); break;
    }
}

// fold hook: the 32 8-fold columns (both forms the kernels use) and the 64 4-fold columns of a scalar
// (ecp_8Folds / ecp_4Folds, source/curve25519_utils.c:144 / :125).  out: 32 + 32 + 64 bytes.
C25519_DEV void fold_selftest_op(uint8_t* out, const u32 (&kw)[8])
{
    u32 k[8];
#pragma unroll
    for (int i = 0; i < 8; i++) k[i] = kw[i];
    for (int n = 0; n < 32; n++) out[n] = (uint8_t)fold8_at(kw, n);
    for (int n = 0; n < 32; n++) out[32 + n] = (uint8_t)fold8_next(k);
#pragma unroll
    for (int i = 0; i < 8; i++) k[i] = kw[i];
    for (int n = 0; n < 64; n++) out[64 + n] = (uint8_t)fold4_next(k, n >= 32);
}

}  // namespace c25519
