// curve25519_amd/csrc/engine.hip -- gfx950 kernels and the C-ABI shim of the batched Curve25519 /
// Ed25519 engine.  One keypair / signature per lane; every arithmetic step of the path runs on the
// device.  Entry points are declared in include/curve25519_amd.h, include/curve25519_dh.h and
// include/ed25519_signature.h (each cites the reference prototype it replaces).
//
// The reference pays one field inversion (ecp_Inverse, 254 S + 11 M) per call (curve25519_dh.c:148,
// ed25519_sign.c:265); here it is shared between several elements with Montgomery's trick:
//   * X25519 is ONE launch (k_x25519_fused): the workgroup's waves park their projective results in LDS and
//     one wave inverts them all;
//   * Ed25519 operations are two or three launches on the caller's stream: a "mult" kernel leaves the
//     projective point in scratch, k_batch_invert (K elements per lane) writes the canonical bytes, and sign
//     adds a finish kernel that hashes enc(R) || pk || m and computes S.
//
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared engine.hip -o libcurve25519_amd.so
#include "capi_common.hpp"
#include "fe25519.cuh"
#include "ge25519.cuh"
#include "sc25519.cuh"
#include "sha512.cuh"
#include "x25519.cuh"

#include "../../include/curve25519_amd.h"
#include "../../include/curve25519_dh.h"
#include "../../include/ed25519_signature.h"

#include <mutex>

using namespace c25519;

#define C25519_RC(expr) do { int rc_ = (expr); if (rc_) return rc_; } while (0)

// ------------------------------------------------------------------------------------------------
// lane I/O
// ------------------------------------------------------------------------------------------------
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

// messages of a batch: fixed stride (offsets == nullptr) or ragged (message i = base[offsets[i] .. offsets[i+1]))
struct Msgs {
    const uint8_t* base;
    size_t fixed;
    const unsigned long long* offsets;
    C25519_DEV const uint8_t* ptr(size_t i) const { return base + (offsets ? (size_t)offsets[i] : i * fixed); }
    C25519_DEV size_t len(size_t i) const { return offsets ? (size_t)(offsets[i + 1] - offsets[i]) : fixed; }
};

// per-call scratch, carved out of one slab (all sizes in u32 words per element)
constexpr size_t SCR_FE = 10;
struct ProjScratch {            // projective result + prefix products of the batched inversion
    u32 *a, *b, *z, *prefix;    // X25519: a = PX, z = PZ.  Edwards: a = X, b = Y, z = Z.
};

// ------------------------------------------------------------------------------------------------
// X25519   (curve25519_dh_CreateSharedKey / curve25519_dh_CalculatePublicKey)
// ------------------------------------------------------------------------------------------------
constexpr int X_BLOCK = 64;

// BASE9 (pk == nullptr): ladder on the base point u = 9
template <bool BASE9>
__global__ void __launch_bounds__(X_BLOCK) k_x25519_ladder(ProjScratch scr, const void* pk, void* sk, size_t n)
{
    const size_t i = (size_t)blockIdx.x * X_BLOCK + threadIdx.x;
    if (i >= n) return;
    u32 u[8] = { 9, 0, 0, 0, 0, 0, 0, 0 }, k[8];
    if (!BASE9) load32(u, pk, i);
    load32(k, sk, i);
    clamp_words(k);
    store32(sk, i, k);                       // the reference clamps in the caller's buffer
    fe PX, PZ;
    x25519_ladder_xz<BASE9>(PX, PZ, u, k);
    soa_store_fe(scr.a, n, i, PX);
    soa_store_fe(scr.z, n, i, PZ);
}

// Single-launch form: the four waves of a workgroup finish their ladders, park (PX, PZ) in LDS, and wave 0
// inverts all the workgroup's Z's with ONE exponentiation (one element per wave and lane, Montgomery's trick, prefix products in
// LDS); then every lane finishes its own element.  Same arithmetic as k_x25519_ladder + k_batch_invert with
// K = 4, but the projective intermediates never leave the CU: HBM traffic is the API's 96 B/op plus the
// clamped-key write-back.
#ifndef C25519_XF_BLOCK
#define C25519_XF_BLOCK 512
#endif
constexpr int XF_BLOCK = C25519_XF_BLOCK;     // waves per workgroup = elements per inverting lane
constexpr int XF_K = XF_BLOCK / 64;

C25519_DEV void lds_put_fe(u32* buf, int stride, int idx, const fe& f)
{
#pragma unroll
    for (int w = 0; w < 10; w++) buf[w * stride + idx] = f.v[w];
}
C25519_DEV void lds_get_fe(fe& f, const u32* buf, int stride, int idx)
{
#pragma unroll
    for (int w = 0; w < 10; w++) f.v[w] = buf[w * stride + idx];
}

template <bool BASE9>
__global__ void __launch_bounds__(XF_BLOCK, 4) k_x25519_fused(void* out, const void* pk, void* sk, size_t n)
{
    __shared__ u32 zbuf[10 * XF_BLOCK];      // PZ, later 1/PZ
    __shared__ u32 xbuf[10 * XF_BLOCK];      // PX
    __shared__ u32 pbuf[(XF_K - 1) * 10 * 64];   // prefix products of the inverting wave
    const int tid = threadIdx.x;
    const size_t i = (size_t)blockIdx.x * XF_BLOCK + tid;
    const bool active = i < n;
    {
        fe PX, PZ;
        if (active) {
            u32 u[8] = { 9, 0, 0, 0, 0, 0, 0, 0 }, k[8];
            if (!BASE9) load32(u, pk, i);
            load32(k, sk, i);
            clamp_words(k);
            store32(sk, i, k);                   // the reference clamps in the caller's buffer
            x25519_ladder_xz<BASE9>(PX, PZ, u, k);
        } else {
            fe_set_u32(PX, 0);
            fe_set_u32(PZ, 1);
        }
        lds_put_fe(zbuf, XF_BLOCK, tid, PZ);
        lds_put_fe(xbuf, XF_BLOCK, tid, PX);
    }
    __syncthreads();
    if (tid < 64) {
        fe acc, z, one, zero;
        fe_set_u32(one, 1);
        fe_set_u32(zero, 0);
        u32 zero_mask = 0;
#pragma unroll 1
        for (int t = 0; t < XF_K; t++) {
            lds_get_fe(z, zbuf, XF_BLOCK, tid + 64 * t);
            u32 w[8], nz = 0;
            fe_to_words(w, z);
#pragma unroll
            for (int q = 0; q < 8; q++) nz |= w[q];
            const u32 is_zero = nz ? 0u : 0xffffffffu;
            zero_mask |= (is_zero & 1u) << t;
            fe_select(z, is_zero, one, z);
            if (t == 0) acc = z; else fe_mul(acc, acc, z);
            if (t < XF_K - 1) lds_put_fe(pbuf + t * 640, 64, tid, acc);
        }
        fe inv;
        fe_invert(inv, acc);
#pragma unroll 1
        for (int t = XF_K - 1; t >= 0; t--) {
            fe zi;
            if (t > 0) {
                fe p;
                lds_get_fe(p, pbuf + (t - 1) * 640, 64, tid);
                fe_mul(zi, inv, p);
                lds_get_fe(z, zbuf, XF_BLOCK, tid + 64 * t);
                const u32 was_zero = ((zero_mask >> t) & 1u) ? 0xffffffffu : 0u;
                fe_select(z, was_zero, one, z);
                fe_mul(inv, inv, z);
                fe_select(zi, was_zero, zero, zi);
            } else {
                const u32 was_zero = (zero_mask & 1u) ? 0xffffffffu : 0u;
                fe_select(zi, was_zero, zero, inv);
            }
            lds_put_fe(zbuf, XF_BLOCK, tid + 64 * t, zi);
        }
    }
    __syncthreads();
    if (active) {
        fe x, zi;
        u32 w[8];
        lds_get_fe(x, xbuf, XF_BLOCK, tid);
        lds_get_fe(zi, zbuf, XF_BLOCK, tid);
        fe_mul(x, x, zi);
        fe_to_words(w, x);
        store32(out, i, w);                      // written last: `out` may alias `pk`
    }
}

// ------------------------------------------------------------------------------------------------
// 8-fold base table, generated on the device at first use
// ------------------------------------------------------------------------------------------------
// row k = sum over set bits i of k of 2^(32 i) * B, as canonical (Y+X, Y-X, 2dT): the content of the
// reference's source/base_folding8.h, derived from B by doubling/adding (the recipe of
// test/curve25519_selftest.c:498-551).  Written twice: limb-major limbs for LDS staging and 96-byte
// canonical rows for inspection.
// Thread group t (256 threads each) produces T_t = 2^((BASE_NT-1-t)*BASE_STEP) * T for the short walk of
// ge_base_mult; the last group is the reference table T itself.
__global__ void __launch_bounds__(256 * BASE_NT) k_gen_base_table(u32* tbl_limbs /*[BASE_NT][30][256]*/,
                                                                  u32* tbl_bytes /*[256][24]*/)
{
    const u32 k = threadIdx.x & 255u;
    const int group = threadIdx.x >> 8;
    const int extra = (BASE_NT - 1 - group) * BASE_STEP;   // trailing doublings
    const bool shifted = extra != 0;
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
    fe zi, x, y, t;
    fe_invert(zi, S.Z);
    fe_mul(x, S.X, zi);
    fe_mul(y, S.Y, zi);
    fe row[3];
    fe_add(row[0], y, x);
    fe_sub(row[1], y, x);
    fe_mul(t, x, y);
    fe_mul(row[2], t, fe_const(K_2D));
    u32* limbs = tbl_limbs + group * BASE_TBL_WORDS;
#pragma unroll
    for (int f = 0; f < 3; f++) {
        u32 w[8];
        fe_to_words(w, row[f]);
        fe c;
        fe_from_words(c, w);                  // canonical value back in limb form
#pragma unroll
        for (int l = 0; l < 10; l++) limbs[(10 * f + l) * 256 + k] = c.v[l];
        if (!shifted) {
#pragma unroll
            for (int j = 0; j < 8; j++) tbl_bytes[k * 24 + 8 * f + j] = w[j];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Ed25519
// ------------------------------------------------------------------------------------------------
constexpr int ED_BLOCK = 256;
constexpr int BM_BLOCK = 1024;            // fixed-base kernels: one 120 KiB table set per 16 waves (4 per SIMD)

// a = clamp(first half of SHA-512(seed)), optionally the second half as 4 big-endian stream words
C25519_DEV void ed_expand_seed(u32 (&a)[8], u64 (&b_words)[4], const u32 (&seed)[8])
{
    u64 pre[4], dg[8];
    sha512_words_from_le32(pre, seed);
    sha512_prefixed<4>(dg, pre, nullptr, 0);
    u32 le[16];
    sha512_digest_le_words(le, dg);
#pragma unroll
    for (int i = 0; i < 8; i++) a[i] = le[i];
    clamp_words(a);
#pragma unroll
    for (int i = 0; i < 4; i++) b_words[i] = dg[4 + i];
}

C25519_DEV void store_proj(const ProjScratch& scr, size_t n, size_t i, const ge_ext& S)
{
    soa_store_fe(scr.a, n, i, S.X);
    soa_store_fe(scr.b, n, i, S.Y);
    soa_store_fe(scr.z, n, i, S.Z);
}

// ed25519_CreateKeyPair (ed25519_sign.c:344-367), first part: a = clamp(H(sk)), S = a*B projective;
// privKey[0..31] = sk.  The public key bytes are written by k_batch_invert<FinishKeypair>.
__global__ void __launch_bounds__(BM_BLOCK, 4) k_ed25519_keypair_mult(ProjScratch scr, void* priv, const void* sk,
                                                                       size_t n, const u32* __restrict__ g_tbl)
{
    __shared__ __attribute__((aligned(16))) u32 lds_tbl[BASE_NT * BASE_TBL_WORDS];
    lds_stage_base_table(lds_tbl, g_tbl, BASE_NT);
    const size_t i = (size_t)blockIdx.x * BM_BLOCK + threadIdx.x;
    if (i >= n) return;
    u32 seed[8], a[8];
    u64 b_words[4];
    load32(seed, sk, i);
    store32(priv, 2 * i, seed);
    ed_expand_seed(a, b_words, seed);
    ge_ext S;
    ge_base_mult(S, a, lds_tbl);
    store_proj(scr, n, i, S);
}

// curve25519_dh_CalculatePublicKey_fast (curve25519_dh.c:162-189): S = clamp(sk)*B, u = (Z+Y)/(Z-Y);
// numerator and denominator go to scratch in the X25519 slots.
__global__ void __launch_bounds__(BM_BLOCK, 4) k_x25519_public_fast_mult(ProjScratch scr, void* sk, size_t n,
                                                                          const u32* __restrict__ g_tbl)
{
    __shared__ __attribute__((aligned(16))) u32 lds_tbl[BASE_NT * BASE_TBL_WORDS];
    lds_stage_base_table(lds_tbl, g_tbl, BASE_NT);
    const size_t i = (size_t)blockIdx.x * BM_BLOCK + threadIdx.x;
    if (i >= n) return;
    u32 k[8];
    load32(k, sk, i);
    clamp_words(k);
    store32(sk, i, k);
    ge_ext S;
    ge_base_mult(S, k, lds_tbl);
    fe num, den, t;
    fe_add(t, S.Z, S.Y);  fe_carry32(num, t);
    fe_sub(t, S.Z, S.Y);  fe_carry32(den, t);
    soa_store_fe(scr.a, n, i, num);
    soa_store_fe(scr.z, n, i, den);
}

// ed25519_SignMessage (ed25519_sign.c:372-419), blinding == NULL, first part (:385-400):
// a = clamp(H(sk)[0..31]), r = H(H(sk)[32..63] || m) mod L (canonical), R = r*B projective.
__global__ void __launch_bounds__(BM_BLOCK, 4) k_ed25519_sign_mult(ProjScratch scr, u32* a_out, u32* r_out,
                                                                    const void* priv, Msgs msgs, size_t n,
                                                                    const u32* __restrict__ g_tbl)
{
    __shared__ __attribute__((aligned(16))) u32 lds_tbl[BASE_NT * BASE_TBL_WORDS];
    lds_stage_base_table(lds_tbl, g_tbl, BASE_NT);
    const size_t i = (size_t)blockIdx.x * BM_BLOCK + threadIdx.x;
    if (i >= n) return;
    u32 seed[8], a[8], r[8];
    {
        u64 b_words[4], dg[8];
        u32 le[16];
        load32(seed, priv, 2 * i);
        ed_expand_seed(a, b_words, seed);
        sha512_prefixed<4>(dg, b_words, msgs.ptr(i), msgs.len(i));
        sha512_digest_le_words(le, dg);
        sc_reduce512(r, le);
        sc_mod(r);
    }
    soa_store8(a_out, n, i, a);
    soa_store8(r_out, n, i, r);
    ge_ext S;
    ge_base_mult(S, r, lds_tbl);
    store_proj(scr, n, i, S);
}

// ... last part (:404-414): h = H(enc(R) || pk || m), S = h*a + r mod L.  enc(R) is already in sig[0..31].
__global__ void __launch_bounds__(ED_BLOCK, 2) k_ed25519_sign_finish(void* sig, const void* priv, Msgs msgs, size_t n,
                                                                      const u32* a_in, const u32* r_in)
{
    const size_t i = (size_t)blockIdx.x * ED_BLOCK + threadIdx.x;
    if (i >= n) return;
    u32 encR[8], pkw[8], a[8], r[8], h[8], s[8], le[16];
    u64 pre[8], dg[8];
    load32(encR, sig, 2 * i);
    load32(pkw, priv, 2 * i + 1);
    sha512_words_from_le32(pre, encR);
    sha512_words_from_le32(pre + 4, pkw);
    sha512_prefixed<8>(dg, pre, msgs.ptr(i), msgs.len(i));
    sha512_digest_le_words(le, dg);
    sc_reduce512(h, le);
    soa_load8(a, a_in, n, i);
    soa_load8(r, r_in, n, i);
    sc_mul(s, h, a);
    sc_add(s, s, r);
    sc_mod(s);
    store32(sig, 2 * i + 1, s);
}

// ed25519_Verify_Init (ed25519_verify.c:179-232): decompress -A (inverted parity :192-195, no validation) and
// fill the key's 16-row 4-fold table.  `tables` holds n tables of Tbl's format, `stride_words` apart.
template <typename Tbl>
__global__ void __launch_bounds__(ED_BLOCK, 2) k_ed25519_verify_init(const void* pk, size_t n, u32* tables,
                                                                      size_t stride_words)
{
    const size_t i = (size_t)blockIdx.x * ED_BLOCK + threadIdx.x;
    if (i >= n) return;
    u32 yw[8];
    load32(yw, pk, i);
    const u32 parity = yw[7] >> 31;
    yw[7] &= 0x7fffffffu;
    ge_ext Q;
    fe_from_words(Q.Y, yw);
    ge_calc_x(Q.X, Q.Y, ~parity);
    fe_mul(Q.T, Q.X, Q.Y);
    fe_set_u32(Q.Z, 1);
    const Tbl tbl{ tables + i * stride_words };
    qtable_build(tbl, Q);
}

// ed25519_Verify_Check (ed25519_verify.c:287-313), first part: h = H(enc(R) || pk || m) mod L canonical;
// s = raw 256 bits (no s < L check, :308); T = s*B + h*(-A) projective.  The comparison with enc(R) happens in
// k_batch_invert<FinishVerify>.   pk_stride 1 = one key per element.
template <typename Tbl>
C25519_DEV void verify_check_lane(const ProjScratch& scr, size_t n, size_t i, const void* sig, const u32 (&pkw)[8],
                                  const Msgs& msgs, const Tbl& tbl, const u32* lds_tbl)
{
    u32 Sw[8], h[8];
    {
        u32 Rw[8], le[16];
        u64 pre[8], dg[8];
        load32(Rw, sig, 2 * i);
        sha512_words_from_le32(pre, Rw);
        sha512_words_from_le32(pre + 4, pkw);
        sha512_prefixed<8>(dg, pre, msgs.ptr(i), msgs.len(i));
        sha512_digest_le_words(le, dg);
        sc_reduce512(h, le);
        sc_mod(h);
    }
    load32(Sw, sig, 2 * i + 1);
    ge_ext T;
    ge_poly_mult(T, Sw, h, tbl, lds_tbl);
    store_proj(scr, n, i, T);
}

template <typename Tbl>
__global__ void __launch_bounds__(ED_BLOCK, 2) k_ed25519_verify_check(ProjScratch scr, const void* sig, const void* pk,
                                                                       Msgs msgs, size_t n,
                                                                       const u32* __restrict__ g_tbl, u32* tables,
                                                                       size_t stride_words)
{
    __shared__ __attribute__((aligned(16))) u32 lds_tbl[PA_WORDS * 256];
    lds_stage_base_table(lds_tbl, g_tbl + (BASE_NT - 1) * BASE_TBL_WORDS);
    const size_t i = (size_t)blockIdx.x * ED_BLOCK + threadIdx.x;
    if (i >= n) return;
    u32 pkw[8];
    load32(pkw, pk, i);
    const Tbl tbl{ tables + i * stride_words };
    verify_check_lane(scr, n, i, sig, pkw, msgs, tbl, lds_tbl);
}

// Same check with ONE key for the whole batch (the reference's two-phase use: Verify_Init once, many
// Verify_Check calls, ed25519_verify.c:282-286).  ctx is the 2080-byte context (pk || 16 canonical rows); the
// workgroup converts it once into limb form in LDS (limb-major, 16 rows wide: the 16 possible row indices
// of a lookup fall into 16 different banks).
struct QTableLds {
    const u32* base;                                       // [40][16]
    C25519_DEV void load(ge_pe& q, u32 e) const
    {
#pragma unroll
        for (int i = 0; i < 10; i++) {
            q.ypx.v[i] = base[(i) * 16 + e];
            q.ymx.v[i] = base[(10 + i) * 16 + e];
            q.t2d.v[i] = base[(20 + i) * 16 + e];
            q.z2.v[i] = base[(30 + i) * 16 + e];
        }
    }
};

__global__ void __launch_bounds__(ED_BLOCK, 2) k_ed25519_verify_check_shared(ProjScratch scr, const void* sig,
                                                                              const u32* __restrict__ ctx, Msgs msgs,
                                                                              size_t n, const u32* __restrict__ g_tbl)
{
    __shared__ __attribute__((aligned(16))) u32 lds_tbl[PA_WORDS * 256];
    __shared__ u32 lds_q[PE_WORDS * 16];
    if (threadIdx.x < 64) {                                // 16 rows x 4 field elements
        const u32 row = threadIdx.x >> 2, f = threadIdx.x & 3;
        u32 w[8];
#pragma unroll
        for (int j = 0; j < 8; j++) w[j] = ctx[8 + row * 32 + f * 8 + j];
        fe v;
        fe_from_words(v, w);
#pragma unroll
        for (int l = 0; l < 10; l++) lds_q[(10 * f + l) * 16 + row] = v.v[l];
    }
    lds_stage_base_table(lds_tbl, g_tbl + (BASE_NT - 1) * BASE_TBL_WORDS);   // ends with __syncthreads()
    const size_t i = (size_t)blockIdx.x * ED_BLOCK + threadIdx.x;
    if (i >= n) return;
    u32 pkw[8];
#pragma unroll
    for (int j = 0; j < 8; j++) pkw[j] = ctx[j];
    const QTableLds tbl{ lds_q };
    verify_check_lane(scr, n, i, sig, pkw, msgs, tbl, lds_tbl);
}

// ------------------------------------------------------------------------------------------------
// batched inversion + output encoding
// ------------------------------------------------------------------------------------------------
// Lane j owns elements j, j+m, j+2m, ... (m = number of lanes, so every access stays coalesced) and inverts
// their Z's with ONE exponentiation: prefix products forward, z^(p-2) once, then unwinding backwards
// (Montgomery's trick).  A zero Z (low-order X25519 input, garbage Ed25519 key) must come out as 0 exactly like
// the reference's z^(p-2) does, so zeros are replaced by 1 in the product and their inverse is forced to 0.
// Fin::emit(e, zinv) turns element e's projective value and 1/Z into the operation's output bytes.
struct FinishX25519 {                       // out = canonical(PX / PZ)           (curve25519_dh.c:148-150)
    const u32* px; void* out; size_t n;
    C25519_DEV void emit(size_t e, const fe& zinv) const
    {
        fe x;
        u32 w[8];
        soa_load_fe(x, px, n, e);
        fe_mul(x, x, zinv);
        fe_to_words(w, x);
        store32(out, e, w);
    }
};

C25519_DEV void affine_pack(u32 (&enc)[8], const u32* X, const u32* Y, size_t n, size_t e, const fe& zinv)
{
    fe t;
    u32 xw[8], yw[8];
    soa_load_fe(t, X, n, e);  fe_mul(t, t, zinv);  fe_to_words(xw, t);     // ed25519_sign.c:265-267
    soa_load_fe(t, Y, n, e);  fe_mul(t, t, zinv);  fe_to_words(yw, t);
    ge_pack(enc, xw, yw);
}

struct FinishPack {                          // 32-byte record `slot` of `stride`-record rows <- enc(x, y)
    const u32 *X, *Y; void* out; size_t n, stride, slot; void* out2; size_t stride2, slot2;
    C25519_DEV void emit(size_t e, const fe& zinv) const
    {
        u32 enc[8];
        affine_pack(enc, X, Y, n, e, zinv);
        store32(out, e * stride + slot, enc);
        if (out2) store32(out2, e * stride2 + slot2, enc);
    }
};

struct FinishVerify {                        // verdict = (enc(T) == enc(R) bytes)   (ed25519_verify.c:310-312)
    const u32 *X, *Y; const void* sig; int* verdict; size_t n;
    C25519_DEV void emit(size_t e, const fe& zinv) const
    {
        u32 enc[8], Rw[8];
        affine_pack(enc, X, Y, n, e, zinv);
        load32(Rw, sig, 2 * e);
        u32 diff = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) diff |= enc[j] ^ Rw[j];
        verdict[e] = diff == 0 ? 1 : 0;
    }
};

constexpr int INV_BLOCK = 64;
constexpr int INV_MAX_K = 16;

template <typename Fin>
__global__ void __launch_bounds__(INV_BLOCK) k_batch_invert(const u32* Z, u32* prefix, size_t n, size_t m, int K, Fin fin)
{
    const size_t j = (size_t)blockIdx.x * INV_BLOCK + threadIdx.x;
    if (j >= m) return;
    fe acc, z;
    u32 zero_mask = 0;
    fe_set_u32(acc, 1);
#pragma unroll 1
    for (int t = 0; t < K; t++) {
        const size_t e = j + (size_t)t * m;
        if (e >= n) break;
        soa_load_fe(z, Z, n, e);
        u32 w[8], nz = 0;
        fe_to_words(w, z);
#pragma unroll
        for (int q = 0; q < 8; q++) nz |= w[q];
        const u32 is_zero = nz ? 0u : 0xffffffffu;
        zero_mask |= (is_zero & 1u) << t;
        fe one;
        fe_set_u32(one, 1);
        fe_select(z, is_zero, one, z);                     // z == 0 (mod p) takes no part in the product
        fe_mul(acc, acc, z);
        soa_store_fe(prefix, n, e, acc);
    }
    fe inv;
    fe_invert(inv, acc);
#pragma unroll 1
    for (int t = K - 1; t >= 0; t--) {
        const size_t e = j + (size_t)t * m;
        if (e >= n) continue;
        fe zi;
        if (t > 0) {
            fe p;
            soa_load_fe(p, prefix, n, e - m);
            fe_mul(zi, inv, p);
        } else {
            zi = inv;
        }
        const u32 was_zero = ((zero_mask >> t) & 1u) ? 0xffffffffu : 0u;
        if (t > 0) {
            soa_load_fe(z, Z, n, e);
            fe one;
            fe_set_u32(one, 1);
            fe_select(z, was_zero, one, z);
            fe_mul(inv, inv, z);
        }
        fe zero;
        fe_set_u32(zero, 0);
        fe_select(zi, was_zero, zero, zi);
        fin.emit(e, zi);
    }
}

// ------------------------------------------------------------------------------------------------
// field-level self-test hook (the counterpart of the reference's ECP_SELF_TEST unit checks,
// test/curve25519_selftest.c:640-741): out[i] = canonical( op(a[i], b[i]) ) on 32-byte little-endian values
// taken mod p.  op: 0 a*b, 1 a^2, 2 a+b, 3 a-b, 4 1/a, 5 a^((p-5)/8), 6 a (canonicalise only),
// 7 (a-b)*(a+b) with unreduced operands (exercises the beta-3 x beta-2 corner of the bound contract).
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) k_fe_selftest(void* out, const void* a, const void* b, size_t n, int op)
{
    const size_t i = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    u32 aw[8], bw[8], ow[8];
    load32(aw, a, i);
    load32(bw, b, i);
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
    default: fe_sub(r, x, y); fe_add(t, x, y); fe_mul(r, r, t); break;
    }
    fe_to_words(ow, r);
    store32(out, i, ow);
}

// ================================================================================================
// host side
// ================================================================================================
namespace {

using c25519_host::Staging;
using c25519_host::aligned16;
using c25519_host::bad_arg;
using c25519_host::staging;

constexpr int MAX_DEVICES = 64;
struct DeviceTables {
    std::once_flag once;
    int rc = 0;
    u32* limbs = nullptr;     // [BASE_NT][30][256]: 2^24 T, 2^16 T, 2^8 T, T
    u32* bytes = nullptr;     // [256][24]
};
DeviceTables g_tables[MAX_DEVICES];

int init_tables(DeviceTables& t)
{
    C25519_TRY(hipMalloc(&t.limbs, BASE_NT * BASE_TBL_WORDS * sizeof(u32)));
    C25519_TRY(hipMalloc(&t.bytes, 256 * 24 * sizeof(u32)));
    k_gen_base_table<<<1, 256 * BASE_NT, 0, nullptr>>>(t.limbs, t.bytes);
    C25519_TRY(hipGetLastError());
    C25519_TRY(hipStreamSynchronize(nullptr));
    return 0;
}

// device-resident 8-fold table of the current device (generated once per device per process)
int base_tables(const u32** limbs, const u32** bytes)
{
    int dev = 0;
    C25519_TRY(hipGetDevice(&dev));
    if (dev < 0 || dev >= MAX_DEVICES) return bad_arg("device ordinal out of range");
    DeviceTables& t = g_tables[dev];
    std::call_once(t.once, [&] { t.rc = init_tables(t); });
    if (t.rc) return t.rc;
    if (limbs) *limbs = t.limbs;
    if (bytes) *bytes = t.bytes;
    return 0;
}

inline unsigned grid_for(size_t n, int block) { return (unsigned)((n + block - 1) / block); }

int check_dev_args(size_t n, std::initializer_list<const void*> ptrs)
{
    if (n > ((size_t)1 << 31)) return bad_arg("batch too large (n > 2^31)");
    for (const void* p : ptrs)
        if (p && !aligned16(p)) return bad_arg("device pointers must be 16-byte aligned");
    return 0;
}

// Work scratch of the *_dev entry points: one grow-only slab per host thread.  Consecutive calls of a thread
// reuse it in stream order; if a thread switches streams, the new stream first waits for the previous use.
struct WorkScratch {
    void* ptr = nullptr;
    size_t cap = 0;
    int dev = -1;
    hipEvent_t done = nullptr;
    hipStream_t last = nullptr;
    bool used = false;

    int acquire(void** out, size_t bytes, hipStream_t stream)
    {
        int d = 0;
        C25519_TRY(hipGetDevice(&d));
        if (ptr && (d != dev || bytes > cap)) {
            C25519_TRY(hipDeviceSynchronize());
            C25519_TRY(hipFree(ptr));
            ptr = nullptr; cap = 0; used = false;
        }
        if (!ptr) {
            size_t want = bytes < ((size_t)1 << 20) ? ((size_t)1 << 20) : bytes;
            C25519_TRY(hipMalloc(&ptr, want));
            cap = want; dev = d;
        }
        if (!done) C25519_TRY(hipEventCreateWithFlags(&done, hipEventDisableTiming));
        if (used && stream != last) C25519_TRY(hipStreamWaitEvent(stream, done, 0));
        *out = ptr;
        return 0;
    }
    int release(hipStream_t stream)
    {
        C25519_TRY(hipEventRecord(done, stream));
        last = stream; used = true;
        return 0;
    }
};
thread_local WorkScratch tl_work;

inline size_t round_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// words of the projective-result part of the scratch for n elements (a, b, z, prefix; 16-byte aligned parts)
inline size_t proj_words(size_t n) { return 4 * round_up(SCR_FE * n, 4); }

ProjScratch carve_proj(u32* base, size_t n)
{
    const size_t part = round_up(SCR_FE * n, 4);
    return ProjScratch{ base, base + part, base + 2 * part, base + 3 * part };
}

// how many elements share one inversion: as many as possible while every SIMD still gets a wave
// (measured at n = 2^20: K = 2 / 4 / 8 / 16 -> 9.52 / 9.39 / 9.33 / 9.29 ms per X25519 pass)
inline int inversion_k(size_t n)
{
    if (const char* e = getenv("C25519_AMD_INV_K")) {      // tuning knob, 1..16
        int v = atoi(e);
        if (v >= 1 && v <= INV_MAX_K) return v;
    }
    size_t k = n / ((size_t)1024 * 64);
    if (k < 1) k = 1;
    if (k > INV_MAX_K) k = INV_MAX_K;
    return (int)k;
}

template <typename Fin>
int launch_invert(const ProjScratch& scr, size_t n, const Fin& fin, hipStream_t stream)
{
    const int K = inversion_k(n);
    const size_t m = (n + K - 1) / K;
    k_batch_invert<Fin><<<grid_for(m, INV_BLOCK), INV_BLOCK, 0, stream>>>(scr.z, scr.prefix, n, m, K, fin);
    C25519_TRY(hipGetLastError());
    return 0;
}

// ---- two-lane host pipeline used by the *_batch entry points ----
struct Lane {                        // one side of the two-deep pipeline
    Staging& s;
    int lane;
    hipStream_t stream() const { return lane ? s.stream2 : s.stream; }
    void* ptr(int slot) const { return s.ptr[4 * lane + slot]; }
    int up(int slot, const void* src, size_t bytes) const
    {
        C25519_RC(s.reserve(4 * lane + slot, bytes));
        if (bytes) C25519_TRY(hipMemcpyAsync(ptr(slot), src, bytes, hipMemcpyHostToDevice, stream()));
        return 0;
    }
    int room(int slot, size_t bytes) const { return s.reserve(4 * lane + slot, bytes); }
    int down(void* dst, int slot, size_t bytes) const
    {
        if (bytes) C25519_TRY(hipMemcpyAsync(dst, ptr(slot), bytes, hipMemcpyDeviceToHost, stream()));
        return 0;
    }
};

// submit(lane, lo, count) uploads a chunk and enqueues its kernels; collect(lane, lo, count) downloads its
// results.  Chunk i+1 is submitted BEFORE chunk i is collected, so its (host-blocking, pageable) upload and
// the download of chunk i both run under the kernels of the neighbouring chunk.
template <typename Submit, typename Collect>
int pipelined(size_t n, Submit submit, Collect collect)
{
    Staging& s = staging();
    C25519_RC(s.ensure_stream());
    const size_t chunk = n >= ((size_t)1 << 18) ? round_up((n + 3) / 4, 64) : n;
    size_t prev_lo = 0, prev_cnt = 0;
    int lane = 0;
    for (size_t lo = 0; lo < n; lo += chunk, lane ^= 1) {
        const size_t cnt = n - lo < chunk ? n - lo : chunk;
        C25519_RC(submit(Lane{ s, lane }, lo, cnt));
        if (prev_cnt) C25519_RC(collect(Lane{ s, lane ^ 1 }, prev_lo, prev_cnt));
        prev_lo = lo; prev_cnt = cnt;
    }
    C25519_RC(collect(Lane{ s, lane ^ 1 }, prev_lo, prev_cnt));
    C25519_TRY(hipStreamSynchronize(s.stream));
    C25519_TRY(hipStreamSynchronize(s.stream2));
    return 0;
}

}  // namespace

extern "C" {

const char* c25519_amd_version(void) { return "curve25519_amd 0.3 (gfx950)"; }
const char* c25519_amd_last_error(void) { return c25519_host::last_error().c_str(); }

int c25519_amd_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}

int c25519_amd_set_device(int device)
{
    C25519_TRY(hipSetDevice(device));
    return 0;
}

// ---- device-pointer entry points ----------------------------------------------------------------

static int x25519_dev(void* out, const void* pk, void* sk, size_t n, hipStream_t stream)
{
    static const bool split = getenv("C25519_AMD_X25519_SPLIT") != nullptr;   // A/B knob: two-launch form
    if (!split) {
        if (pk) k_x25519_fused<false><<<grid_for(n, XF_BLOCK), XF_BLOCK, 0, stream>>>(out, pk, sk, n);
        else    k_x25519_fused<true><<<grid_for(n, XF_BLOCK), XF_BLOCK, 0, stream>>>(out, pk, sk, n);
        C25519_TRY(hipGetLastError());
        return 0;
    }
    void* w = nullptr;
    C25519_RC(tl_work.acquire(&w, proj_words(n) * sizeof(u32), stream));
    const ProjScratch scr = carve_proj((u32*)w, n);
    if (pk) k_x25519_ladder<false><<<grid_for(n, X_BLOCK), X_BLOCK, 0, stream>>>(scr, pk, sk, n);
    else    k_x25519_ladder<true><<<grid_for(n, X_BLOCK), X_BLOCK, 0, stream>>>(scr, pk, sk, n);
    C25519_TRY(hipGetLastError());
    C25519_RC(launch_invert(scr, n, FinishX25519{ scr.a, out, n }, stream));
    return tl_work.release(stream);
}

int curve25519_dh_CreateSharedKey_dev(void* shared, const void* pk, void* sk, size_t n, void* stream)
{
    if (!shared || !pk || !sk) return bad_arg("null pointer");
    if (int rc = check_dev_args(n, { shared, pk, sk })) return rc;
    if (n == 0) return 0;
    return x25519_dev(shared, pk, sk, n, (hipStream_t)stream);
}

int curve25519_dh_CalculatePublicKey_dev(void* pk, void* sk, size_t n, void* stream)
{
    if (!pk || !sk) return bad_arg("null pointer");
    if (int rc = check_dev_args(n, { pk, sk })) return rc;
    if (n == 0) return 0;
    return x25519_dev(pk, nullptr, sk, n, (hipStream_t)stream);
}

int curve25519_dh_CalculatePublicKey_fast_dev(void* pk, void* sk, size_t n, void* stream_)
{
    if (!pk || !sk) return bad_arg("null pointer");
    if (int rc = check_dev_args(n, { pk, sk })) return rc;
    if (n == 0) return 0;
    hipStream_t stream = (hipStream_t)stream_;
    const u32* tbl = nullptr;
    C25519_RC(base_tables(&tbl, nullptr));
    void* w = nullptr;
    C25519_RC(tl_work.acquire(&w, proj_words(n) * sizeof(u32), stream));
    const ProjScratch scr = carve_proj((u32*)w, n);
    k_x25519_public_fast_mult<<<grid_for(n, BM_BLOCK), BM_BLOCK, 0, stream>>>(scr, sk, n, tbl);
    C25519_TRY(hipGetLastError());
    C25519_RC(launch_invert(scr, n, FinishX25519{ scr.a, pk, n }, stream));
    return tl_work.release(stream);
}

int ed25519_CreateKeyPair_dev(void* pub, void* priv, const void* sk, size_t n, void* stream_)
{
    if (!pub || !priv || !sk) return bad_arg("null pointer");
    if (int rc = check_dev_args(n, { pub, priv, sk })) return rc;
    if (n == 0) return 0;
    hipStream_t stream = (hipStream_t)stream_;
    const u32* tbl = nullptr;
    C25519_RC(base_tables(&tbl, nullptr));
    void* w = nullptr;
    C25519_RC(tl_work.acquire(&w, proj_words(n) * sizeof(u32), stream));
    const ProjScratch scr = carve_proj((u32*)w, n);
    k_ed25519_keypair_mult<<<grid_for(n, BM_BLOCK), BM_BLOCK, 0, stream>>>(scr, priv, sk, n, tbl);
    C25519_TRY(hipGetLastError());
    // pub[e] and priv[e][32..63] <- enc(A)
    C25519_RC(launch_invert(scr, n, FinishPack{ scr.a, scr.b, pub, n, 1, 0, priv, 2, 1 }, stream));
    return tl_work.release(stream);
}

static int sign_dev(void* sig, const void* priv, Msgs msgs, size_t n, hipStream_t stream)
{
    if (int rc = check_dev_args(n, { sig, priv })) return rc;
    if (n == 0) return 0;
    const u32* tbl = nullptr;
    C25519_RC(base_tables(&tbl, nullptr));
    void* w = nullptr;
    const size_t sc_words = round_up(8 * n, 4);
    C25519_RC(tl_work.acquire(&w, (proj_words(n) + 2 * sc_words) * sizeof(u32), stream));
    const ProjScratch scr = carve_proj((u32*)w, n);
    u32* a_buf = (u32*)w + proj_words(n);
    u32* r_buf = a_buf + sc_words;
    k_ed25519_sign_mult<<<grid_for(n, BM_BLOCK), BM_BLOCK, 0, stream>>>(scr, a_buf, r_buf, priv, msgs, n, tbl);
    C25519_TRY(hipGetLastError());
    C25519_RC(launch_invert(scr, n, FinishPack{ scr.a, scr.b, sig, n, 2, 0, nullptr, 0, 0 }, stream));   // sig[e][0..31] = enc(R)
    k_ed25519_sign_finish<<<grid_for(n, ED_BLOCK), ED_BLOCK, 0, stream>>>(sig, priv, msgs, n, a_buf, r_buf);
    C25519_TRY(hipGetLastError());
    return tl_work.release(stream);
}

int ed25519_SignMessage_dev(void* sig, const void* priv, const void* msg, size_t msg_size, size_t n, void* stream)
{
    if (!sig || !priv || (!msg && msg_size)) return bad_arg("null pointer");
    return sign_dev(sig, priv, Msgs{ (const uint8_t*)msg, msg_size, nullptr }, n, (hipStream_t)stream);
}

int ed25519_SignMessage_ragged_dev(void* sig, const void* priv, const void* msgs, const uint64_t* offsets, size_t n,
                                   void* stream)
{
    if (!sig || !priv || !offsets) return bad_arg("null pointer");
    return sign_dev(sig, priv, Msgs{ (const uint8_t*)msgs, 0, (const unsigned long long*)offsets }, n, (hipStream_t)stream);
}

size_t ed25519_VerifySignature_scratch_bytes(size_t n)
{
    return (n * QTABLE_LIMB_WORDS + proj_words(n)) * sizeof(u32);
}

static int verify_dev(void* verdict, const void* sig, const void* pk, Msgs msgs, size_t n, hipStream_t stream)
{
    if (int rc = check_dev_args(n, { sig, pk })) return rc;
    if (n == 0) return 0;
    const u32* tbl = nullptr;
    C25519_RC(base_tables(&tbl, nullptr));
    void* w = nullptr;
    C25519_RC(tl_work.acquire(&w, ed25519_VerifySignature_scratch_bytes(n), stream));
    const ProjScratch scr = carve_proj((u32*)w, n);
    u32* tables = (u32*)w + proj_words(n);
    k_ed25519_verify_init<QTableLimbs><<<grid_for(n, ED_BLOCK), ED_BLOCK, 0, stream>>>(pk, n, tables, QTABLE_LIMB_WORDS);
    C25519_TRY(hipGetLastError());
    k_ed25519_verify_check<QTableLimbs><<<grid_for(n, ED_BLOCK), ED_BLOCK, 0, stream>>>(
        scr, sig, pk, msgs, n, tbl, tables, QTABLE_LIMB_WORDS);
    C25519_TRY(hipGetLastError());
    C25519_RC(launch_invert(scr, n, FinishVerify{ scr.a, scr.b, sig, (int*)verdict, n }, stream));
    return tl_work.release(stream);
}

// test hook: enc(T) instead of the verdict (what Verify_Check compares with enc(R)); device pointers
int c25519_amd_verify_point_dev(void* out, const void* sig, const void* pk, const void* msg, size_t msg_size, size_t n,
                                void* stream_)
{
    if (!out || !sig || !pk || (!msg && msg_size)) return bad_arg("null pointer");
    if (int rc = check_dev_args(n, { out, sig, pk })) return rc;
    if (n == 0) return 0;
    hipStream_t stream = (hipStream_t)stream_;
    const u32* tbl = nullptr;
    C25519_RC(base_tables(&tbl, nullptr));
    void* w = nullptr;
    C25519_RC(tl_work.acquire(&w, ed25519_VerifySignature_scratch_bytes(n), stream));
    const ProjScratch scr = carve_proj((u32*)w, n);
    u32* tables = (u32*)w + proj_words(n);
    const Msgs msgs{ (const uint8_t*)msg, msg_size, nullptr };
    k_ed25519_verify_init<QTableLimbs><<<grid_for(n, ED_BLOCK), ED_BLOCK, 0, stream>>>(pk, n, tables, QTABLE_LIMB_WORDS);
    C25519_TRY(hipGetLastError());
    k_ed25519_verify_check<QTableLimbs><<<grid_for(n, ED_BLOCK), ED_BLOCK, 0, stream>>>(
        scr, sig, pk, msgs, n, tbl, tables, QTABLE_LIMB_WORDS);
    C25519_TRY(hipGetLastError());
    C25519_RC(launch_invert(scr, n, FinishPack{ scr.a, scr.b, out, n, 1, 0, nullptr, 0, 0 }, stream));
    return tl_work.release(stream);
}

int ed25519_VerifySignature_dev(void* verdict, const void* sig, const void* pk, const void* msg, size_t msg_size,
                                size_t n, void* stream)
{
    if (!verdict || !sig || !pk || (!msg && msg_size)) return bad_arg("null pointer");
    return verify_dev(verdict, sig, pk, Msgs{ (const uint8_t*)msg, msg_size, nullptr }, n, (hipStream_t)stream);
}

int ed25519_VerifySignature_ragged_dev(void* verdict, const void* sig, const void* pk, const void* msgs,
                                       const uint64_t* offsets, size_t n, void* stream)
{
    if (!verdict || !sig || !pk || !offsets) return bad_arg("null pointer");
    return verify_dev(verdict, sig, pk, Msgs{ (const uint8_t*)msgs, 0, (const unsigned long long*)offsets }, n,
                      (hipStream_t)stream);
}

// two-phase verification on the device: contexts are 2080-byte records (pk || 16 x 128-byte canonical rows),
// the reference's EDP_SIGV_CTX size and row order.
int ed25519_Verify_Init_dev(void* ctx, const void* pk, size_t n, void* stream)
{
    if (!ctx || !pk) return bad_arg("null pointer");
    if (int rc = check_dev_args(n, { ctx, pk })) return rc;
    if (n == 0) return 0;
    C25519_TRY(hipMemcpy2DAsync(ctx, 2080, pk, 32, 32, n, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    k_ed25519_verify_init<QTableCanon><<<grid_for(n, ED_BLOCK), ED_BLOCK, 0, (hipStream_t)stream>>>(
        pk, n, (u32*)ctx + 8, 2080 / 4);
    C25519_TRY(hipGetLastError());
    return 0;
}

int ed25519_Verify_Check_dev(void* verdict, const void* ctx, const void* sig, const void* msg, size_t msg_size,
                             size_t n, void* stream_)
{
    if (!verdict || !ctx || !sig || (!msg && msg_size)) return bad_arg("null pointer");
    if (int rc = check_dev_args(n, { ctx, sig })) return rc;
    if (n == 0) return 0;
    hipStream_t stream = (hipStream_t)stream_;
    const u32* tbl = nullptr;
    C25519_RC(base_tables(&tbl, nullptr));
    void* w = nullptr;
    C25519_RC(tl_work.acquire(&w, proj_words(n) * sizeof(u32), stream));
    const ProjScratch scr = carve_proj((u32*)w, n);
    k_ed25519_verify_check_shared<<<grid_for(n, ED_BLOCK), ED_BLOCK, 0, stream>>>(
        scr, sig, (const u32*)ctx, Msgs{ (const uint8_t*)msg, msg_size, nullptr }, n, tbl);
    C25519_TRY(hipGetLastError());
    C25519_RC(launch_invert(scr, n, FinishVerify{ scr.a, scr.b, sig, (int*)verdict, n }, stream));
    return tl_work.release(stream);
}

int c25519_amd_fe_selftest(unsigned char* out, const unsigned char* a, const unsigned char* b, size_t n, int op)
{
    if (!out || !a || !b) return bad_arg("null pointer");
    if (n == 0) return 0;
    Staging& s = staging();
    C25519_RC(s.ensure_stream());
    C25519_RC(s.reserve(0, 32 * n));
    C25519_RC(s.reserve(1, 32 * n));
    C25519_RC(s.reserve(2, 32 * n));
    C25519_TRY(hipMemcpyAsync(s.ptr[0], a, 32 * n, hipMemcpyHostToDevice, s.stream));
    C25519_TRY(hipMemcpyAsync(s.ptr[1], b, 32 * n, hipMemcpyHostToDevice, s.stream));
    k_fe_selftest<<<grid_for(n, 64), 64, 0, s.stream>>>(s.ptr[2], s.ptr[0], s.ptr[1], n, op);
    C25519_TRY(hipGetLastError());
    C25519_TRY(hipMemcpyAsync(out, s.ptr[2], 32 * n, hipMemcpyDeviceToHost, s.stream));
    C25519_TRY(hipStreamSynchronize(s.stream));
    return 0;
}

int c25519_amd_base_table(unsigned char* out)
{
    if (!out) return bad_arg("null pointer");
    const u32* bytes = nullptr;
    if (int rc = base_tables(nullptr, &bytes)) return rc;
    C25519_TRY(hipMemcpy(out, bytes, 256 * 96, hipMemcpyDeviceToHost));
    return 0;
}

// ---- host-pointer entry points: stage, run the *_dev form, copy back -----------------------------
// Large batches are cut into four chunks that alternate between two (stream, staging-slot) lanes, so the
// PCIe copies of one chunk run under the kernels of the other.  Pageable host memory: the copy blocks the
// calling thread, not the GPU.

// legacy single-lane helpers (ragged entry points)
static int up(Staging& s, int slot, const void* src, size_t bytes)
{
    C25519_RC(s.reserve(slot, bytes));
    if (bytes) C25519_TRY(hipMemcpyAsync(s.ptr[slot], src, bytes, hipMemcpyHostToDevice, s.stream));
    return 0;
}
static int down(Staging& s, void* dst, int slot, size_t bytes)
{
    if (bytes) C25519_TRY(hipMemcpyAsync(dst, s.ptr[slot], bytes, hipMemcpyDeviceToHost, s.stream));
    return 0;
}

int curve25519_dh_CreateSharedKey_batch(unsigned char* shared, const unsigned char* pk, unsigned char* sk, size_t n)
{
    if (!shared || !pk || !sk) return bad_arg("null pointer");
    if (n == 0) return 0;
    return pipelined(n,
        [&](const Lane& L, size_t lo, size_t c) -> int {
            C25519_RC(L.up(0, pk + 32 * lo, 32 * c));
            C25519_RC(L.up(1, sk + 32 * lo, 32 * c));
            C25519_RC(L.room(2, 32 * c));
            return curve25519_dh_CreateSharedKey_dev(L.ptr(2), L.ptr(0), L.ptr(1), c, L.stream());
        },
        [&](const Lane& L, size_t lo, size_t c) -> int {
            C25519_RC(L.down(sk + 32 * lo, 1, 32 * c));
            return L.down(shared + 32 * lo, 2, 32 * c);
        });
}

static int public_batch(unsigned char* pk, unsigned char* sk, size_t n, bool fast)
{
    if (!pk || !sk) return bad_arg("null pointer");
    if (n == 0) return 0;
    return pipelined(n,
        [&](const Lane& L, size_t lo, size_t c) -> int {
            C25519_RC(L.up(1, sk + 32 * lo, 32 * c));
            C25519_RC(L.room(2, 32 * c));
            return fast ? curve25519_dh_CalculatePublicKey_fast_dev(L.ptr(2), L.ptr(1), c, L.stream())
                        : curve25519_dh_CalculatePublicKey_dev(L.ptr(2), L.ptr(1), c, L.stream());
        },
        [&](const Lane& L, size_t lo, size_t c) -> int {
            C25519_RC(L.down(sk + 32 * lo, 1, 32 * c));
            return L.down(pk + 32 * lo, 2, 32 * c);
        });
}

int curve25519_dh_CalculatePublicKey_batch(unsigned char* pk, unsigned char* sk, size_t n) { return public_batch(pk, sk, n, false); }
int curve25519_dh_CalculatePublicKey_fast_batch(unsigned char* pk, unsigned char* sk, size_t n) { return public_batch(pk, sk, n, true); }

int ed25519_CreateKeyPair_batch(unsigned char* pub, unsigned char* priv, const unsigned char* sk, size_t n)
{
    if (!pub || !priv || !sk) return bad_arg("null pointer");
    if (n == 0) return 0;
    return pipelined(n,
        [&](const Lane& L, size_t lo, size_t c) -> int {
            C25519_RC(L.up(0, sk + 32 * lo, 32 * c));
            C25519_RC(L.room(1, 32 * c));
            C25519_RC(L.room(2, 64 * c));
            return ed25519_CreateKeyPair_dev(L.ptr(1), L.ptr(2), L.ptr(0), c, L.stream());
        },
        [&](const Lane& L, size_t lo, size_t c) -> int {
            C25519_RC(L.down(pub + 32 * lo, 1, 32 * c));
            return L.down(priv + 64 * lo, 2, 64 * c);
        });
}

int ed25519_SignMessage_batch(unsigned char* sig, const unsigned char* priv, const unsigned char* msg,
                              size_t msg_size, size_t n)
{
    if (!sig || !priv || (!msg && msg_size)) return bad_arg("null pointer");
    if (n == 0) return 0;
    return pipelined(n,
        [&](const Lane& L, size_t lo, size_t c) -> int {
            C25519_RC(L.up(0, priv + 64 * lo, 64 * c));
            C25519_RC(L.up(1, msg ? msg + msg_size * lo : nullptr, msg_size * c));
            C25519_RC(L.room(2, 64 * c));
            return ed25519_SignMessage_dev(L.ptr(2), L.ptr(0), L.ptr(1), msg_size, c, L.stream());
        },
        [&](const Lane& L, size_t lo, size_t c) -> int { return L.down(sig + 64 * lo, 2, 64 * c); });
}

int ed25519_VerifySignature_batch(int* verdict, const unsigned char* sig, const unsigned char* pk,
                                  const unsigned char* msg, size_t msg_size, size_t n)
{
    if (!verdict || !sig || !pk || (!msg && msg_size)) return bad_arg("null pointer");
    if (n == 0) return 0;
    return pipelined(n,
        [&](const Lane& L, size_t lo, size_t c) -> int {
            C25519_RC(L.up(0, sig + 64 * lo, 64 * c));
            C25519_RC(L.up(1, pk + 32 * lo, 32 * c));
            C25519_RC(L.up(2, msg ? msg + msg_size * lo : nullptr, msg_size * c));
            C25519_RC(L.room(3, sizeof(int) * c));
            return ed25519_VerifySignature_dev(L.ptr(3), L.ptr(0), L.ptr(1), L.ptr(2), msg_size, c, L.stream());
        },
        [&](const Lane& L, size_t lo, size_t c) -> int { return L.down(verdict + lo, 3, sizeof(int) * c); });
}

// ragged messages: message i is msgs[offsets[i] .. offsets[i+1]); offsets has n+1 entries (host memory)
int ed25519_SignMessage_ragged_batch(unsigned char* sig, const unsigned char* priv, const unsigned char* msgs,
                                     const uint64_t* offsets, size_t n)
{
    if (!sig || !priv || !offsets) return bad_arg("null pointer");
    if (n == 0) return 0;
    Staging& s = staging();
    C25519_RC(s.ensure_stream());
    C25519_RC(up(s, 0, priv, 64 * n));
    C25519_RC(up(s, 1, msgs, (size_t)offsets[n]));
    C25519_RC(s.reserve(2, 64 * n));
    C25519_RC(up(s, 3, offsets, sizeof(uint64_t) * (n + 1)));
    C25519_RC(ed25519_SignMessage_ragged_dev(s.ptr[2], s.ptr[0], s.ptr[1], (const uint64_t*)s.ptr[3], n, s.stream));
    C25519_RC(down(s, sig, 2, 64 * n));
    C25519_TRY(hipStreamSynchronize(s.stream));
    return 0;
}

int ed25519_VerifySignature_ragged_batch(int* verdict, const unsigned char* sig, const unsigned char* pk,
                                         const unsigned char* msgs, const uint64_t* offsets, size_t n)
{
    if (!verdict || !sig || !pk || !offsets) return bad_arg("null pointer");
    if (n == 0) return 0;
    Staging& s = staging();
    C25519_RC(s.ensure_stream());
    C25519_RC(up(s, 0, sig, 64 * n));
    C25519_RC(up(s, 1, pk, 32 * n));
    C25519_RC(up(s, 2, msgs, (size_t)offsets[n]));
    C25519_RC(s.reserve(3, sizeof(int) * n));
    C25519_RC(up(s, 4, offsets, sizeof(uint64_t) * (n + 1)));
    C25519_RC(ed25519_VerifySignature_ragged_dev(s.ptr[3], s.ptr[0], s.ptr[1], s.ptr[2], (const uint64_t*)s.ptr[4], n,
                                                 s.stream));
    C25519_RC(down(s, verdict, 3, sizeof(int) * n));
    C25519_TRY(hipStreamSynchronize(s.stream));
    return 0;
}

// ---- the reference's single-call API: a device batch of one, fatal on device failure --------------

void curve25519_dh_CalculatePublicKey(unsigned char* pk, unsigned char* sk)
{
    if (int rc = curve25519_dh_CalculatePublicKey_batch(pk, sk, 1)) c25519_host::die(__func__, rc);
}

void curve25519_dh_CalculatePublicKey_fast(unsigned char* pk, unsigned char* sk)
{
    if (int rc = curve25519_dh_CalculatePublicKey_fast_batch(pk, sk, 1)) c25519_host::die(__func__, rc);
}

void curve25519_dh_CreateSharedKey(unsigned char* shared, const unsigned char* pk, unsigned char* sk)
{
    if (int rc = curve25519_dh_CreateSharedKey_batch(shared, pk, sk, 1)) c25519_host::die(__func__, rc);
}

void ed25519_CreateKeyPair(unsigned char* pubKey, unsigned char* privKey, const void* blinding, const unsigned char* sk)
{
    (void)blinding;                           // output-neutral in the reference (ed25519_sign.c:254-263)
    if (int rc = ed25519_CreateKeyPair_batch(pubKey, privKey, sk, 1)) c25519_host::die(__func__, rc);
}

void ed25519_SignMessage(unsigned char* signature, const unsigned char* privKey, const void* blinding,
                         const unsigned char* msg, size_t msg_size)
{
    (void)blinding;
    if (int rc = ed25519_SignMessage_batch(signature, privKey, msg, msg_size, 1)) c25519_host::die(__func__, rc);
}

int ed25519_VerifySignature(const unsigned char* signature, const unsigned char* publicKey, const unsigned char* msg,
                            size_t msg_size)
{
    int verdict = 0;
    if (int rc = ed25519_VerifySignature_batch(&verdict, signature, publicKey, msg, msg_size, 1))
        c25519_host::die(__func__, rc);
    return verdict;
}

// Blinding contexts: accepted and carried for API compatibility; they hold the caller's seed digest
// position only (no arithmetic depends on them, matching the reference's observable behaviour).
struct blinding_ctx { unsigned char opaque[192]; };

void* ed25519_Blinding_Init(void* context, const unsigned char* seed, size_t size)
{
    blinding_ctx* ctx = (blinding_ctx*)context;
    if (!ctx) {
        ctx = (blinding_ctx*)malloc(sizeof(blinding_ctx));
        if (!ctx) return nullptr;
    }
    memset(ctx->opaque, 0, sizeof ctx->opaque);
    for (size_t i = 0; i < size; i++) ctx->opaque[i % sizeof ctx->opaque] ^= seed[i];
    return ctx;
}

void ed25519_Blinding_Finish(void* context)
{
    if (context) {
        memset(context, 0, sizeof(blinding_ctx));
        free(context);
    }
}

// Two-phase verification.  The context is the reference's EDP_SIGV_CTX shape (2080 bytes: pk, then 16
// rows of four canonical field elements), filled by the device; it lives in the caller's storage or is
// malloc'ed here, exactly as in the reference (ed25519_verify.c:179-237).
int ed25519_Verify_Init_batch(void* ctx, const unsigned char* pk, size_t n)
{
    if (!ctx || !pk) return bad_arg("null pointer");
    if (n == 0) return 0;
    Staging& s = staging();
    C25519_RC(s.ensure_stream());
    C25519_RC(up(s, 0, pk, 32 * n));
    C25519_RC(s.reserve(1, 2080 * n));
    C25519_RC(ed25519_Verify_Init_dev(s.ptr[1], s.ptr[0], n, s.stream));
    C25519_RC(down(s, ctx, 1, 2080 * n));
    C25519_TRY(hipStreamSynchronize(s.stream));
    return 0;
}

int ed25519_Verify_Check_batch(int* verdict, const void* ctx, const unsigned char* sig, const unsigned char* msg,
                               size_t msg_size, size_t n)
{
    if (!verdict || !ctx || !sig || (!msg && msg_size)) return bad_arg("null pointer");
    if (n == 0) return 0;
    Staging& s = staging();
    C25519_RC(s.ensure_stream());
    C25519_RC(up(s, 0, sig, 64 * n));
    C25519_RC(up(s, 1, ctx, 2080));
    C25519_RC(up(s, 2, msg, msg_size * n));
    C25519_RC(s.reserve(3, sizeof(int) * n));
    C25519_RC(ed25519_Verify_Check_dev(s.ptr[3], s.ptr[1], s.ptr[0], s.ptr[2], msg_size, n, s.stream));
    C25519_RC(down(s, verdict, 3, sizeof(int) * n));
    C25519_TRY(hipStreamSynchronize(s.stream));
    return 0;
}

void* ed25519_Verify_Init(void* context, const unsigned char* publicKey)
{
    void* ctx = context ? context : malloc(2080);
    if (!ctx) return nullptr;                  // allocation failure is the only error the reference reports
    if (int rc = ed25519_Verify_Init_batch(ctx, publicKey, 1)) c25519_host::die(__func__, rc);
    return ctx;
}

int ed25519_Verify_Check(const void* context, const unsigned char* signature, const unsigned char* msg, size_t msg_size)
{
    int verdict = 0;
    if (int rc = ed25519_Verify_Check_batch(&verdict, context, signature, msg, msg_size, 1))
        c25519_host::die(__func__, rc);
    return verdict;
}

void ed25519_Verify_Finish(void* ctx) { free(ctx); }

}  // extern "C"
