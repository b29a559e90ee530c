// curve25519_amd/csrc/engine.hip -- gfx950 kernels and the C-ABI shim of the batched Curve25519 /
// Ed25519 engine.  One keypair / signature per lane; every arithmetic step of the path runs on the
// device.  Entry points are declared in include/curve25519_amd.h, include/curve25519_dh.h and
// include/ed25519_signature.h (each cites the reference prototype it replaces).
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

// ------------------------------------------------------------------------------------------------
// lane I/O: 32-byte records as two 16-byte accesses (a wave covers 2 KiB of contiguous memory)
// ------------------------------------------------------------------------------------------------
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

// ------------------------------------------------------------------------------------------------
// X25519   (curve25519_dh_CreateSharedKey / curve25519_dh_CalculatePublicKey)
// ------------------------------------------------------------------------------------------------
constexpr int X_BLOCK = 64;

// pk == nullptr: base point u = 9
__global__ void __launch_bounds__(X_BLOCK) k_x25519(void* out, const void* pk, void* sk, size_t n)
{
    const size_t i = (size_t)blockIdx.x * X_BLOCK + threadIdx.x;
    if (i >= n) return;
    u32 u[8] = { 9, 0, 0, 0, 0, 0, 0, 0 }, k[8], o[8];
    if (pk) load32(u, pk, i);
    load32(k, sk, i);
    clamp_words(k);
    store32(sk, i, k);                       // the reference clamps in the caller's buffer
    x25519_ladder(o, u, k);
    store32(out, i, o);                      // written last: `out` may alias `pk`
}

// ------------------------------------------------------------------------------------------------
// 8-fold base table, generated on the device at first use
// ------------------------------------------------------------------------------------------------
// row k = sum over set bits i of k of 2^(32 i) * B, as canonical (Y+X, Y-X, 2dT): the content of the
// reference's source/base_folding8.h, derived from B by doubling/adding (the recipe of
// test/curve25519_selftest.c:498-551).  Written twice: limb-major limbs for LDS staging and 96-byte
// canonical rows for inspection.
__global__ void __launch_bounds__(256) k_gen_base_table(u32* tbl_limbs /*[30][256]*/, u32* tbl_bytes /*[256][24]*/)
{
    const u32 k = threadIdx.x;
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
        if (i) {
#pragma unroll 1
            for (int j = 0; j < 32; j++) ge_double(S);
        }
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
#pragma unroll
    for (int f = 0; f < 3; f++) {
        u32 w[8];
        fe_to_words(w, row[f]);
        fe c;
        fe_from_words(c, w);                  // canonical value back in limb form
#pragma unroll
        for (int l = 0; l < 10; l++) tbl_limbs[(10 * f + l) * 256 + k] = c.v[l];
#pragma unroll
        for (int j = 0; j < 8; j++) tbl_bytes[k * 24 + 8 * f + j] = w[j];
    }
}

// ------------------------------------------------------------------------------------------------
// Ed25519
// ------------------------------------------------------------------------------------------------
constexpr int ED_BLOCK = 256;

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

// packed canonical encoding of k*B
C25519_DEV void ed_base_mult_packed(u32 (&enc)[8], u32 (&k)[8], const u32* lds_tbl)
{
    ge_ext S;
    u32 xw[8], yw[8];
    ge_base_mult(S, k, lds_tbl);
    ge_to_affine_words(xw, yw, S);
    ge_pack(enc, xw, yw);
}

// ed25519_CreateKeyPair (ed25519_sign.c:344-367): pub = enc(a*B), priv = sk || pub
__global__ void __launch_bounds__(ED_BLOCK) k_ed25519_keypair(void* pub, void* priv, const void* sk, size_t n,
                                                               const u32* __restrict__ g_tbl)
{
    __shared__ __attribute__((aligned(16))) u32 lds_tbl[PA_WORDS * 256];
    lds_stage_base_table(lds_tbl, g_tbl);
    const size_t i = (size_t)blockIdx.x * ED_BLOCK + threadIdx.x;
    if (i >= n) return;
    u32 seed[8], a[8], enc[8];
    u64 b_words[4];
    load32(seed, sk, i);
    ed_expand_seed(a, b_words, seed);
    ed_base_mult_packed(enc, a, lds_tbl);
    store32(pub, i, enc);
    store32(priv, 2 * i, seed);
    store32(priv, 2 * i + 1, enc);
}

// curve25519_dh_CalculatePublicKey_fast (curve25519_dh.c:162-189): u = (Z+Y)/(Z-Y) of clamp(sk)*B
__global__ void __launch_bounds__(ED_BLOCK) k_x25519_public_fast(void* pk, void* sk, size_t n,
                                                                  const u32* __restrict__ g_tbl)
{
    __shared__ __attribute__((aligned(16))) u32 lds_tbl[PA_WORDS * 256];
    lds_stage_base_table(lds_tbl, g_tbl);
    const size_t i = (size_t)blockIdx.x * ED_BLOCK + threadIdx.x;
    if (i >= n) return;
    u32 k[8], o[8];
    load32(k, sk, i);
    clamp_words(k);
    store32(sk, i, k);
    ge_ext S;
    ge_base_mult(S, k, lds_tbl);
    fe num, den, t;
    fe_add(num, S.Z, S.Y);                    // beta 2
    fe_sub(t, S.Z, S.Y);                      // beta 3
    fe_carry32(den, t);
    fe_invert(den, den);
    fe_mul(t, num, den);
    fe_to_words(o, t);
    store32(pk, i, o);
}

// ed25519_SignMessage (ed25519_sign.c:372-419), blinding == NULL
__global__ void __launch_bounds__(ED_BLOCK) k_ed25519_sign(void* sig, const void* priv, const uint8_t* msg,
                                                            size_t msg_size, size_t n, const u32* __restrict__ g_tbl)
{
    __shared__ __attribute__((aligned(16))) u32 lds_tbl[PA_WORDS * 256];
    lds_stage_base_table(lds_tbl, g_tbl);
    const size_t i = (size_t)blockIdx.x * ED_BLOCK + threadIdx.x;
    if (i >= n) return;
    const uint8_t* m = msg + i * msg_size;

    u32 seed[8], pkw[8], a[8];
    u64 b_words[4], dg[8];
    load32(seed, priv, 2 * i);
    load32(pkw, priv, 2 * i + 1);
    ed_expand_seed(a, b_words, seed);         // a = clamp(H(sk)[0..31]), b = H(sk)[32..63]   (:385-389)

    u32 r[8], le[16];                         // r = H(b || m) mod L, canonical            (:392-397)
    sha512_prefixed<4>(dg, b_words, m, msg_size);
    sha512_digest_le_words(le, dg);
    sc_reduce512(r, le);
    sc_mod(r);

    u32 rk[8], encR[8];                       // R = r*B                                    (:400-401)
#pragma unroll
    for (int j = 0; j < 8; j++) rk[j] = r[j];
    ed_base_mult_packed(encR, rk, lds_tbl);

    u64 pre[8];                               // h = H(enc(R) || pk || m)                    (:404-409)
    sha512_words_from_le32(pre, encR);
    sha512_words_from_le32(pre + 4, pkw);
    sha512_prefixed<8>(dg, pre, m, msg_size);
    sha512_digest_le_words(le, dg);
    u32 h[8], s[8];
    sc_reduce512(h, le);
    sc_mul(s, h, a);                          // S = h*a + r mod L                            (:411-413)
    sc_add(s, s, r);
    sc_mod(s);

    store32(sig, 2 * i, encR);
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

// ed25519_Verify_Check (ed25519_verify.c:287-313): h = H(enc(R) || pk || m) mod L canonical; s = raw 256
// bits (no s < L check, :308); T = s*B + h*(-A); verdict = (enc(T) == enc(R)).
// pk_stride / table stride 0 = one key for the whole batch (two-phase API), otherwise one key per element.
template <typename Tbl>
__global__ void __launch_bounds__(ED_BLOCK, 2) k_ed25519_verify_check(int* verdict, const void* sig, const void* pk,
                                                                       size_t pk_stride, const uint8_t* msg,
                                                                       size_t msg_size, size_t n,
                                                                       const u32* __restrict__ g_tbl, u32* tables,
                                                                       size_t stride_words)
{
    __shared__ __attribute__((aligned(16))) u32 lds_tbl[PA_WORDS * 256];
    lds_stage_base_table(lds_tbl, g_tbl);
    const size_t i = (size_t)blockIdx.x * ED_BLOCK + threadIdx.x;
    if (i >= n) return;

    u32 Rw[8], Sw[8], h[8];
    load32(Rw, sig, 2 * i);
    load32(Sw, sig, 2 * i + 1);
    {
        u32 pkw[8], le[16];
        u64 pre[8], dg[8];
        load32(pkw, pk, i * pk_stride);
        sha512_words_from_le32(pre, Rw);
        sha512_words_from_le32(pre + 4, pkw);
        sha512_prefixed<8>(dg, pre, msg + i * msg_size, msg_size);
        sha512_digest_le_words(le, dg);
        sc_reduce512(h, le);
        sc_mod(h);
    }

    const Tbl tbl{ tables + i * stride_words };
    ge_ext T;
    ge_poly_mult(T, Sw, h, tbl, lds_tbl);
    u32 xw[8], yw[8], enc[8];
    ge_to_affine_words(xw, yw, T);
    ge_pack(enc, xw, yw);
    u32 diff = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) diff |= enc[j] ^ Rw[j];
    verdict[i] = diff == 0 ? 1 : 0;           // memcmp(md, signature, 32) == 0   (:312)
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

__global__ void __launch_bounds__(ED_BLOCK, 2) k_ed25519_verify_check_shared(int* verdict, const void* sig,
                                                                              const u32* __restrict__ ctx,
                                                                              const uint8_t* msg, size_t msg_size,
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
    lds_stage_base_table(lds_tbl, g_tbl);                  // ends with __syncthreads()
    const size_t i = (size_t)blockIdx.x * ED_BLOCK + threadIdx.x;
    if (i >= n) return;

    u32 Rw[8], Sw[8], h[8];
    load32(Rw, sig, 2 * i);
    load32(Sw, sig, 2 * i + 1);
    {
        u32 pkw[8], le[16];
        u64 pre[8], dg[8];
#pragma unroll
        for (int j = 0; j < 8; j++) pkw[j] = ctx[j];
        sha512_words_from_le32(pre, Rw);
        sha512_words_from_le32(pre + 4, pkw);
        sha512_prefixed<8>(dg, pre, msg + i * msg_size, msg_size);
        sha512_digest_le_words(le, dg);
        sc_reduce512(h, le);
        sc_mod(h);
    }
    const QTableLds tbl{ lds_q };
    ge_ext T;
    ge_poly_mult(T, Sw, h, tbl, lds_tbl);
    u32 xw[8], yw[8], enc[8];
    ge_to_affine_words(xw, yw, T);
    ge_pack(enc, xw, yw);
    u32 diff = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) diff |= enc[j] ^ Rw[j];
    verdict[i] = diff == 0 ? 1 : 0;
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
    u32* limbs = nullptr;     // [30][256]
    u32* bytes = nullptr;     // [256][24]
};
DeviceTables g_tables[MAX_DEVICES];
thread_local size_t tl_qscratch_cap = 0;
thread_local void* tl_qscratch = nullptr;
thread_local int tl_qscratch_dev = -1;

int init_tables(DeviceTables& t)
{
    C25519_TRY(hipMalloc(&t.limbs, PA_WORDS * 256 * sizeof(u32)));
    C25519_TRY(hipMalloc(&t.bytes, 256 * 24 * sizeof(u32)));
    k_gen_base_table<<<1, 256, 0, nullptr>>>(t.limbs, t.bytes);
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

int verify_scratch(void** out, size_t n)
{
    const size_t need = ed25519_VerifySignature_scratch_bytes(n);
    int dev = 0;
    C25519_TRY(hipGetDevice(&dev));
    if (tl_qscratch && (dev != tl_qscratch_dev || need > tl_qscratch_cap)) {
        C25519_TRY(hipFree(tl_qscratch));
        tl_qscratch = nullptr; tl_qscratch_cap = 0;
    }
    if (!tl_qscratch) {
        C25519_TRY(hipMalloc(&tl_qscratch, need));
        tl_qscratch_cap = need; tl_qscratch_dev = dev;
    }
    *out = tl_qscratch;
    return 0;
}

}  // namespace

extern "C" {

const char* c25519_amd_version(void) { return "curve25519_amd 0.1 (gfx950)"; }
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

int curve25519_dh_CreateSharedKey_dev(void* shared, const void* pk, void* sk, size_t n, void* stream)
{
    if (!shared || !pk || !sk) return bad_arg("null pointer");
    if (int rc = check_dev_args(n, { shared, pk, sk })) return rc;
    if (n == 0) return 0;
    k_x25519<<<grid_for(n, X_BLOCK), X_BLOCK, 0, (hipStream_t)stream>>>(shared, pk, sk, n);
    C25519_TRY(hipGetLastError());
    return 0;
}

int curve25519_dh_CalculatePublicKey_dev(void* pk, void* sk, size_t n, void* stream)
{
    if (!pk || !sk) return bad_arg("null pointer");
    if (int rc = check_dev_args(n, { pk, sk })) return rc;
    if (n == 0) return 0;
    k_x25519<<<grid_for(n, X_BLOCK), X_BLOCK, 0, (hipStream_t)stream>>>(pk, nullptr, sk, n);
    C25519_TRY(hipGetLastError());
    return 0;
}

int curve25519_dh_CalculatePublicKey_fast_dev(void* pk, void* sk, size_t n, void* stream)
{
    if (!pk || !sk) return bad_arg("null pointer");
    if (int rc = check_dev_args(n, { pk, sk })) return rc;
    if (n == 0) return 0;
    const u32* tbl = nullptr;
    if (int rc = base_tables(&tbl, nullptr)) return rc;
    k_x25519_public_fast<<<grid_for(n, ED_BLOCK), ED_BLOCK, 0, (hipStream_t)stream>>>(pk, sk, n, tbl);
    C25519_TRY(hipGetLastError());
    return 0;
}

int ed25519_CreateKeyPair_dev(void* pub, void* priv, const void* sk, size_t n, void* stream)
{
    if (!pub || !priv || !sk) return bad_arg("null pointer");
    if (int rc = check_dev_args(n, { pub, priv, sk })) return rc;
    if (n == 0) return 0;
    const u32* tbl = nullptr;
    if (int rc = base_tables(&tbl, nullptr)) return rc;
    k_ed25519_keypair<<<grid_for(n, ED_BLOCK), ED_BLOCK, 0, (hipStream_t)stream>>>(pub, priv, sk, n, tbl);
    C25519_TRY(hipGetLastError());
    return 0;
}

int ed25519_SignMessage_dev(void* sig, const void* priv, const void* msg, size_t msg_size, size_t n, void* stream)
{
    if (!sig || !priv || (!msg && msg_size)) return bad_arg("null pointer");
    if (int rc = check_dev_args(n, { sig, priv })) return rc;
    if (n == 0) return 0;
    const u32* tbl = nullptr;
    if (int rc = base_tables(&tbl, nullptr)) return rc;
    k_ed25519_sign<<<grid_for(n, ED_BLOCK), ED_BLOCK, 0, (hipStream_t)stream>>>(
        sig, priv, (const uint8_t*)msg, msg_size, n, tbl);
    C25519_TRY(hipGetLastError());
    return 0;
}

size_t ed25519_VerifySignature_scratch_bytes(size_t n)
{
    return n * QTABLE_LIMB_WORDS * sizeof(u32);
}

int ed25519_VerifySignature_dev(void* verdict, const void* sig, const void* pk, const void* msg, size_t msg_size,
                                size_t n, void* stream)
{
    if (!verdict || !sig || !pk || (!msg && msg_size)) return bad_arg("null pointer");
    if (int rc = check_dev_args(n, { sig, pk })) return rc;
    if (n == 0) return 0;
    const u32* tbl = nullptr;
    if (int rc = base_tables(&tbl, nullptr)) return rc;
    void* scratch = nullptr;
    if (int rc = verify_scratch(&scratch, n)) return rc;
    k_ed25519_verify_init<QTableLimbs><<<grid_for(n, ED_BLOCK), ED_BLOCK, 0, (hipStream_t)stream>>>(
        pk, n, (u32*)scratch, QTABLE_LIMB_WORDS);
    C25519_TRY(hipGetLastError());
    k_ed25519_verify_check<QTableLimbs><<<grid_for(n, ED_BLOCK), ED_BLOCK, 0, (hipStream_t)stream>>>(
        (int*)verdict, sig, pk, 1, (const uint8_t*)msg, msg_size, n, tbl, (u32*)scratch, QTABLE_LIMB_WORDS);
    C25519_TRY(hipGetLastError());
    return 0;
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
                             size_t n, void* stream)
{
    if (!verdict || !ctx || !sig || (!msg && msg_size)) return bad_arg("null pointer");
    if (int rc = check_dev_args(n, { ctx, sig })) return rc;
    if (n == 0) return 0;
    const u32* tbl = nullptr;
    if (int rc = base_tables(&tbl, nullptr)) return rc;
    k_ed25519_verify_check_shared<<<grid_for(n, ED_BLOCK), ED_BLOCK, 0, (hipStream_t)stream>>>(
        (int*)verdict, sig, (const u32*)ctx, (const uint8_t*)msg, msg_size, n, tbl);
    C25519_TRY(hipGetLastError());
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

#define C25519_RC(expr) do { int rc_ = (expr); if (rc_) return rc_; } while (0)

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
    Staging& s = staging();
    C25519_RC(s.ensure_stream());
    C25519_RC(up(s, 0, pk, 32 * n));
    C25519_RC(up(s, 1, sk, 32 * n));
    C25519_RC(s.reserve(2, 32 * n));
    C25519_RC(curve25519_dh_CreateSharedKey_dev(s.ptr[2], s.ptr[0], s.ptr[1], n, s.stream));
    C25519_RC(down(s, sk, 1, 32 * n));
    C25519_RC(down(s, shared, 2, 32 * n));
    C25519_TRY(hipStreamSynchronize(s.stream));
    return 0;
}

static int public_batch(unsigned char* pk, unsigned char* sk, size_t n, bool fast)
{
    if (!pk || !sk) return bad_arg("null pointer");
    if (n == 0) return 0;
    Staging& s = staging();
    C25519_RC(s.ensure_stream());
    C25519_RC(up(s, 1, sk, 32 * n));
    C25519_RC(s.reserve(2, 32 * n));
    if (fast) C25519_RC(curve25519_dh_CalculatePublicKey_fast_dev(s.ptr[2], s.ptr[1], n, s.stream));
    else      C25519_RC(curve25519_dh_CalculatePublicKey_dev(s.ptr[2], s.ptr[1], n, s.stream));
    C25519_RC(down(s, sk, 1, 32 * n));
    C25519_RC(down(s, pk, 2, 32 * n));
    C25519_TRY(hipStreamSynchronize(s.stream));
    return 0;
}

int curve25519_dh_CalculatePublicKey_batch(unsigned char* pk, unsigned char* sk, size_t n) { return public_batch(pk, sk, n, false); }
int curve25519_dh_CalculatePublicKey_fast_batch(unsigned char* pk, unsigned char* sk, size_t n) { return public_batch(pk, sk, n, true); }

int ed25519_CreateKeyPair_batch(unsigned char* pub, unsigned char* priv, const unsigned char* sk, size_t n)
{
    if (!pub || !priv || !sk) return bad_arg("null pointer");
    if (n == 0) return 0;
    Staging& s = staging();
    C25519_RC(s.ensure_stream());
    C25519_RC(up(s, 0, sk, 32 * n));
    C25519_RC(s.reserve(1, 32 * n));
    C25519_RC(s.reserve(2, 64 * n));
    C25519_RC(ed25519_CreateKeyPair_dev(s.ptr[1], s.ptr[2], s.ptr[0], n, s.stream));
    C25519_RC(down(s, pub, 1, 32 * n));
    C25519_RC(down(s, priv, 2, 64 * n));
    C25519_TRY(hipStreamSynchronize(s.stream));
    return 0;
}

int ed25519_SignMessage_batch(unsigned char* sig, const unsigned char* priv, const unsigned char* msg,
                              size_t msg_size, size_t n)
{
    if (!sig || !priv || (!msg && msg_size)) return bad_arg("null pointer");
    if (n == 0) return 0;
    Staging& s = staging();
    C25519_RC(s.ensure_stream());
    C25519_RC(up(s, 0, priv, 64 * n));
    C25519_RC(up(s, 1, msg, msg_size * n));
    C25519_RC(s.reserve(2, 64 * n));
    C25519_RC(ed25519_SignMessage_dev(s.ptr[2], s.ptr[0], s.ptr[1], msg_size, n, s.stream));
    C25519_RC(down(s, sig, 2, 64 * n));
    C25519_TRY(hipStreamSynchronize(s.stream));
    return 0;
}

int ed25519_VerifySignature_batch(int* verdict, const unsigned char* sig, const unsigned char* pk,
                                  const unsigned char* msg, size_t msg_size, size_t n)
{
    if (!verdict || !sig || !pk || (!msg && msg_size)) return bad_arg("null pointer");
    if (n == 0) return 0;
    Staging& s = staging();
    C25519_RC(s.ensure_stream());
    C25519_RC(up(s, 0, sig, 64 * n));
    C25519_RC(up(s, 1, pk, 32 * n));
    C25519_RC(up(s, 2, msg, msg_size * n));
    C25519_RC(s.reserve(3, sizeof(int) * n));
    C25519_RC(ed25519_VerifySignature_dev(s.ptr[3], s.ptr[0], s.ptr[1], s.ptr[2], msg_size, n, s.stream));
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
