// curve25519_amd/csrc/engine_fixed_base.hip -- the constant tables (generated on the device at first use) and the fixed-base operations: key pairs,
// signatures, curve25519_dh_CalculatePublicKey_fast, blinding contexts -- kernels and *_dev entry points
// (one of the engine's four translation units: engine_common.cuh says which is which)
#include "engine_common.cuh"

// ------------------------------------------------------------------------------------------------
// 8-fold base table, generated on the device at first use
// ------------------------------------------------------------------------------------------------
// Workgroup t < BASE_NT (128 threads each): the signed comb table T_t = 2^((BASE_NT-1-t)*BASE_STEP) * Ts of
// ge_base_mult (ge_signed_comb_row).  Two more workgroups: row k = sum over set bits i of k of 2^(32 i) * B as canonical
// (Y+X, Y-X, 2dT) -- the content of the reference's source/base_folding8.h, derived from B by doubling/adding (the recipe
// of test/curve25519_selftest.c:498-551) -- written twice: limb-major limbs after the signed tables (REF_TBL_OFFSET:
// the reference-order verification's sigma columns) and 96-byte canonical rows for inspection.  SC_ROWS / 128 more: the
// lattice walk's signed comb table (SC_TBL_OFFSET).
__global__ void __launch_bounds__(BASE_ROWS) k_gen_base_table(u32* tbl_limbs /*[BASE_NT][30][128] + [30][256] + [30][SC_ROWS]*/,
                                                              u32* tbl_bytes /*[256][24]*/)
{
    u32 rows[3][8];
    if (blockIdx.x < BASE_NT) {                               // workgroup g: signed comb table g, one row per thread
        const u32 idx = threadIdx.x;
        const int group = blockIdx.x;
        ge_signed_comb_row(rows, idx, (BASE_NT - 1 - group) * BASE_STEP);
        u32* limbs = tbl_limbs + group * BASE_TBL_WORDS;
#pragma unroll
        for (int f = 0; f < 3; f++) {
            fe c;
            fe_from_words(c, rows[f]);            // canonical value back in limb form
#pragma unroll
            for (int l = 0; l < 10; l++) limbs[(10 * f + l) * BASE_ROWS + idx] = c.v[l];
        }
        return;
    }
    if (blockIdx.x >= BASE_NT + 256 / BASE_ROWS) {            // the verification walk's signed comb: SC_ROWS rows
        const u32 idx = (blockIdx.x - (BASE_NT + 256 / BASE_ROWS)) * BASE_ROWS + threadIdx.x;
        ge_signed_comb_row(rows, idx, 0, SC_TEETH, SC_COLS);
        u32* limbs = tbl_limbs + SC_TBL_OFFSET;
#pragma unroll
        for (int f = 0; f < 3; f++) {
            fe c;
            fe_from_words(c, rows[f]);
#pragma unroll
            for (int l = 0; l < 10; l++) limbs[(10 * f + l) * SC_ROWS + idx] = c.v[l];
        }
        return;
    }
    const u32 k = (blockIdx.x - BASE_NT) * BASE_ROWS + threadIdx.x;   // two more workgroups: the reference table's 256 rows
    ge_base_table_row(rows, k, 0);
    u32* limbs = tbl_limbs + REF_TBL_OFFSET;
#pragma unroll
    for (int f = 0; f < 3; f++) {
        fe c;
        fe_from_words(c, rows[f]);
#pragma unroll
        for (int l = 0; l < 10; l++) limbs[(10 * f + l) * 256 + k] = c.v[l];
#pragma unroll
        for (int j = 0; j < 8; j++) tbl_bytes[k * 24 + 8 * f + j] = rows[f][j];
    }
}

// the wide comb's WB_NT tables (ge25519.cuh): one packed 128-byte row per thread, generated on first use of BASE_COMB = 1
__global__ void __launch_bounds__(128) k_gen_wide_table(u32* wide /*[WB_NT][WB_ROWS][WB_ROW_WORDS]*/)
{
    const u32 g = blockIdx.x * 128 + threadIdx.x;             // table * WB_ROWS + row
    const int table = (int)(g / WB_ROWS);
    u32 rows[3][8];
    ge_signed_comb_row(rows, g % WB_ROWS, (WB_NT - 1 - table) * WB_STEP, WB_TEETH, WB_COLS);
    uint4* out = reinterpret_cast<uint4*>(wide + (size_t)g * WB_ROW_WORDS);
#pragma unroll
    for (int f = 0; f < 3; f++) {
        out[2 * f] = make_uint4(rows[f][0], rows[f][1], rows[f][2], rows[f][3]);
        out[2 * f + 1] = make_uint4(rows[f][4], rows[f][5], rows[f][6], rows[f][7]);
    }
    out[6] = make_uint4(2, 0, 0, 0);                          // the row's fourth field: 2Z of an affine point (quad25519.cuh reads a row
    out[7] = make_uint4(0, 0, 0, 0);                          // as the four factors of an addition, one per lane)
}

// ------------------------------------------------------------------------------------------------
// Ed25519
// ------------------------------------------------------------------------------------------------
constexpr int BM_BLOCK = 1024;            // fixed-base kernels: one 120 KiB set of signed comb tables per 16 waves (4 per SIMD)
// ... for batches that fill the chip.  The tables allow one workgroup per CU whatever its size, so a small batch runs
// narrower workgroups on more CUs: 2^14 elements are 16 workgroups of 1024 (16 CUs, four waves per SIMD) or 64 of 256 (one
// wave per SIMD), which come back sooner (profiles/r03_batch_sweep.txt).
// (a piece of a pipelined *_batch call takes the shape of the whole call: host_pipeline.hpp, batch_shape_hint)
inline unsigned bm_block_for(size_t n)
{
    n = std::max(n, c25519_host::batch_shape_hint());
    return n <= ((size_t)1 << 16) ? 256u : n <= ((size_t)1 << 17) ? 512u : (unsigned)BM_BLOCK;
}

// The fixed-base kernels come in two shapes (tunable BASE_COMB, A/B: profiles/r05_ab_base_comb.txt):
//   WIDE = false  the 8 x 32 signed comb, eight tables staged in 120 KiB of LDS per 1024-lane workgroup: 31 additions + 3 doublings;
//   WIDE = true   the 13 x 20 signed comb of ge25519.cuh read through L2: 19 additions + 4 doublings, 256-lane workgroups, the
//                 only LDS the lanes' parked column numbers (10 KiB).
template <bool WIDE> struct BaseComb;
template <> struct BaseComb<false> {
    static constexpr int BLOCK = BM_BLOCK;
    u32* lds;
    C25519_DEV void stage(const u32* __restrict__ g_tbl) const { lds_stage_words(lds, g_tbl, BASE_NT * BASE_TBL_WORDS); }
    template <bool BLIND>
    C25519_DEV void mult(ge_ext& S, const u32 (&k)[8], const u32* __restrict__, const u32* blind_ctx) const
    {
        if (BLIND) ge_base_mult_blinded(S, k, blind_ctx, lds);
        else ge_base_mult(S, k, lds);
    }
};
template <> struct BaseComb<true> {
    static constexpr int BLOCK = WB_BLOCK;
    unsigned short* cols;                                     // [WB_COLS][blockDim.x]
    C25519_DEV void stage(const u32* __restrict__) const {}
    template <bool BLIND>
    C25519_DEV void mult(ge_ext& S, const u32 (&k)[8], const u32* __restrict__ g_wide, const u32* blind_ctx) const
    {
        unsigned short* mine = cols + threadIdx.x;
        const int stride = (int)blockDim.x;
        if (BLIND) {
            ge_base_mult_blinded_with(S, k, blind_ctx, [&](ge_ext& P, const u32 (&t)[8], const fe& zr) {
                wb_columns(mine, stride, t);
                ge_base_mult_wide<true>(P, g_wide, mine, stride, &zr);
            });
        } else {
            wb_columns(mine, stride, k);
            ge_base_mult_wide(S, g_wide, mine, stride);
        }
    }
};
#define C25519_BASE_COMB_SETUP(comb)                                                                          \
    __shared__ __attribute__((aligned(16))) u32 comb##_lds[WIDE ? WB_COLS * WB_BLOCK / 2 : BASE_NT * BASE_TBL_WORDS]; \
    BaseComb<WIDE> comb;                                                                                      \
    if constexpr (WIDE) comb.cols = reinterpret_cast<unsigned short*>(comb##_lds); else comb.lds = comb##_lds; \
    comb.stage(g_tbl)

// ed25519_CreateKeyPair (ed25519_sign.c:344-367), first part: a = clamp(H(sk)), S = a*B projective;
// privKey[0..31] = sk.  The public key bytes are written by k_batch_invert<FinishPack>.
// (g_tbl: the LDS comb's tables in device memory, or the wide comb's)
template <bool BLIND, bool WIDE>
__global__ void __launch_bounds__(BaseComb<WIDE>::BLOCK, 4) k_ed25519_keypair_mult(ProjScratch scr, void* priv, const void* sk,
                                                                                    size_t n, const u32* __restrict__ g_tbl,
                                                                                    const u32* __restrict__ blind_ctx)
{
    C25519_BASE_COMB_SETUP(comb);
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u32 seed[8], a[8];
    u64 b_words[4];
    load32(seed, sk, i);
    store32(priv, 2 * i, seed);
    ed_expand_seed(a, b_words, seed);
    ge_ext S;
    comb.template mult<BLIND>(S, a, g_tbl, blind_ctx);
    store_proj(scr, n, i, S);
}

// curve25519_dh_CalculatePublicKey_fast (curve25519_dh.c:162-189): S = clamp(sk)*B, u = (Z+Y)/(Z-Y);
// numerator and denominator go to scratch in the X25519 slots.
template <bool WIDE>
__global__ void __launch_bounds__(BaseComb<WIDE>::BLOCK, 4) k_x25519_public_fast_mult(ProjScratch scr, void* sk, size_t n,
                                                                                       const u32* __restrict__ g_tbl)
{
    C25519_BASE_COMB_SETUP(comb);
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u32 k[8];
    load32(k, sk, i);
    clamp_words(k);
    store32(sk, i, k);
    ge_ext S;
    comb.template mult<false>(S, k, g_tbl, nullptr);
    fe num, den, t;
    fe_add(t, S.Z, S.Y);  fe_carry32(num, t);
    fe_sub(t, S.Z, S.Y);  fe_carry32(den, t);
    soa_store_fe(scr.a, n, i, num);
    soa_store_fe(scr.z, n, i, den);
}

// ed25519_SignMessage (ed25519_sign.c:372-419), first part (:385-400):
// a = clamp(H(sk)[0..31]), r = H(H(sk)[32..63] || m) mod L (canonical), R = r*B projective.
template <bool BLIND, bool WIDE>
__global__ void __launch_bounds__(BaseComb<WIDE>::BLOCK, 4) k_ed25519_sign_mult(ProjScratch scr, u32* a_out, u32* r_out,
                                                                                 const void* priv, Msgs msgs, size_t n,
                                                                                 const u32* __restrict__ g_tbl,
                                                                                 const u32* __restrict__ blind_ctx)
{
    C25519_BASE_COMB_SETUP(comb);
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u32 seed[8], a[8], r[8];
    load32(seed, priv, 2 * i);
    ed_sign_nonce(a, r, seed, msgs.ptr(i), msgs.len(i));
    soa_store8(a_out, n, i, a);
    soa_store8(r_out, n, i, r);
    ge_ext S;
    comb.template mult<BLIND>(S, r, g_tbl, blind_ctx);
    store_proj(scr, n, i, S);
}

// ... last part (:404-414): h = H(enc(R) || pk || m), S = h*a + r mod L.  enc(R) is already in sig[0..31].
// The scratch copies of a and r are zeroed behind the read (the reference clears its a and r, :416-417).
__global__ void __launch_bounds__(ED_BLOCK, 2) k_ed25519_sign_finish(void* sig, const void* priv, Msgs msgs, size_t n,
                                                                      u32* a_in, u32* r_in)
{
    const size_t i = (size_t)blockIdx.x * ED_BLOCK + threadIdx.x;
    if (i >= n) return;
    u32 encR[8], pkw[8], a[8], r[8], s[8];
    const u32 zero[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
    load32(encR, sig, 2 * i);
    load32(pkw, priv, 2 * i + 1);
    soa_load8(a, a_in, n, i);
    soa_load8(r, r_in, n, i);
    soa_store8(a_in, n, i, zero);
    soa_store8(r_in, n, i, zero);
    ed_sign_s(s, encR, pkw, msgs.ptr(i), msgs.len(i), a, r);
    store32(sig, 2 * i + 1, s);
}

// The same three operations for a call of a few elements, ONE operation per wave (coop25519.cuh): hashing and scalar
// arithmetic by every lane on the same values, the fixed-base walk, the inversion and the affine conversion cooperative.
// (A blinding context: over the wide comb only -- with the LDS comb a blinded call runs the batch kernels.)
// Signatures: a workgroup of TWO waves -- the second computes the message schedules of the three hashes' SHA-512 compressions a
// chunk ahead of the first wave's rounds (coop25519.cuh: ShaTwoWaves) and is gone when the last hash is: one signature 65.8 ->
// 63.0 us.  (Measured from inside, profiles/r06_sign_stamps.txt: a compression 7.0 -> 5.5 us, not the 4.3 its instruction count
// promised -- the rounds alone are one dependent chain, which the schedule's instructions used to fill; a key pair's single block
// gains nothing over the second wave's launch and stays on one wave.)
constexpr int COOP_SHA_BLOCK = 128;
template <bool WIDE>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 4)))
k_ed25519_keypair_coop(void* pub, void* priv, const void* sk, size_t n, const u32* __restrict__ g_tbl,
                       const u32* __restrict__ blind_ctx, DoneWord done, CallWords cw)
{
    __shared__ __attribute__((aligned(16))) u32 lds[coop::LDS_WORDS];
    __shared__ __attribute__((aligned(16))) u32 inl[CALL_WORDS];
    if (blockIdx.x >= n) return;
    if (cw.use) sk = stage_call_words(inl, cw);            // a call of one: the secret came with the arguments (no PCIe read)
    coop::keypair_one<WIDE>(lds, coop::make_lane(threadIdx.x), pub, priv, sk, blockIdx.x, g_tbl, blind_ctx, &done);
}

// curve25519_dh_CalculatePublicKey_fast (curve25519_dh.c:162-189): S = clamp(sk) * B on the Edwards side, u = (Z + Y) / (Z - Y)
template <bool WIDE>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 4)))
k_x25519_public_fast_coop(void* pk, void* sk, size_t n, const u32* __restrict__ g_tbl, DoneWord done)
{
    __shared__ __attribute__((aligned(16))) u32 lds[coop::LDS_WORDS];
    if (blockIdx.x >= n) return;
    coop::public_fast_one<WIDE>(lds, coop::make_lane(threadIdx.x), pk, sk, blockIdx.x, g_tbl, &done);
}

template <bool WIDE>
__global__ void __launch_bounds__(COOP_SHA_BLOCK) __attribute__((amdgpu_waves_per_eu(1, 4)))
k_ed25519_sign_coop(void* sig, const void* priv, Msgs msgs, size_t n, const u32* __restrict__ g_tbl,
                    const u32* __restrict__ blind_ctx, DoneWord done, CallWords cw)
{
    __shared__ __attribute__((aligned(16))) u32 lds[coop::LDS_WORDS];
    __shared__ __attribute__((aligned(16))) u32 inl[CALL_WORDS];
    __shared__ u64 sha_wk[80];
    if (blockIdx.x >= n) return;
    if (cw.use) {                                          // a call of one: key and message came with the arguments (three PCIe reads less)
        priv = stage_call_words(inl, cw);
        msgs.base = reinterpret_cast<const uint8_t*>(inl + 16);
    }
    if (threadIdx.x >= 64) {                               // H(sk), H(prefix || m), H(enc(R) || pk || m)
        const size_t len = msgs.len(blockIdx.x);
        coop::sha_schedule_server(sha_wk, 1 + sha512_blocks(4, len) + sha512_blocks(8, len));
        return;
    }
    coop::sign_one<WIDE>(lds, coop::make_lane(threadIdx.x), sig, priv, msgs, blockIdx.x, g_tbl, blind_ctx, &done, coop::ShaTwoWaves{ sha_wk });
}

// The same three operations on FOUR lanes per element (quad25519.cuh), for calls between the per-wave kernels and the batches
// that fill the chip: 16 elements per one-wave workgroup, the walk over the wide comb in two product levels per addition,
// inversion, encoding and the last hash in the same launch (the one-lane path's three launches are 80 + 56 + 15 us for 2^12 ..
// 2^14 signatures whatever their number; this is one chain of ~110 us).  LDS: the lanes' parked column numbers.
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 2)))
k_ed25519_keypair_quad(void* pub, void* priv, const void* sk, size_t n, const u32* __restrict__ g_wide)
{
    __shared__ unsigned short cols[WB_COLS * 64];
    const size_t e = (size_t)blockIdx.x * quad::ELEMS_PER_WAVE + (threadIdx.x >> 2);
    if (e >= n) return;                                       // (whole quads leave)
    quad::keypair_element(pub, priv, sk, e, g_wide, cols + threadIdx.x, 64);
}

__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 2)))
k_x25519_public_fast_quad(void* pk, void* sk, size_t n, const u32* __restrict__ g_wide)
{
    __shared__ unsigned short cols[WB_COLS * 64];
    const size_t e = (size_t)blockIdx.x * quad::ELEMS_PER_WAVE + (threadIdx.x >> 2);
    if (e >= n) return;
    quad::public_fast_element(pk, sk, e, g_wide, cols + threadIdx.x, 64);
}

__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 2)))
k_ed25519_sign_quad(void* sig, const void* priv, Msgs msgs, size_t n, const u32* __restrict__ g_wide)
{
    __shared__ unsigned short cols[WB_COLS * 64];
    const size_t e = (size_t)blockIdx.x * quad::ELEMS_PER_WAVE + (threadIdx.x >> 2);
    if (e >= n) return;
    quad::sign_element(sig, priv, msgs.ptr(e), msgs.len(e), e, g_wide, cols + threadIdx.x, 64);
}

// ed25519_Blinding_Init (ed25519_sign.c:289-331) for one context: digest = SHA-512(domain || seed),
// t = digest[0..31] mod L, bl = L - t, zr = digest[32..63], BP = PE(t*B).  One lane does the arithmetic; the
// workgroup only stages the base tables.  The domain string replaces the reference's compiled-in custom blinder
// (custom_blind.c), which likewise only seeds the derivation.
__global__ void __launch_bounds__(256) k_ed25519_blinding_init(u32* ctx, const uint8_t* seed, size_t seed_len,
                                                                const u32* __restrict__ g_tbl)
{
    __shared__ __attribute__((aligned(16))) u32 lds_tbl[BASE_NT * BASE_TBL_WORDS];
    lds_stage_words(lds_tbl, g_tbl, BASE_NT * BASE_TBL_WORDS);
    if (threadIdx.x == 0) ed_blinding_init_lane(ctx, seed, seed_len, lds_tbl);
}

// ... and with the whole wave (the wide comb's rows fetched from device memory, t * B and its affine conversion cooperative): what
// ed25519_Blinding_Init runs unless the LDS comb is selected -- 196 -> ~70 us per context
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 4)))
k_ed25519_blinding_init_coop(u32* ctx, const uint8_t* seed, size_t seed_len, const u32* __restrict__ wide, DoneWord done)
{
    __shared__ __attribute__((aligned(16))) u32 lds[coop::LDS_WORDS];
    coop::blinding_init_one(lds, coop::make_lane(threadIdx.x), ctx, seed, seed_len, wide, &done);
}

namespace {

constexpr int MAX_DEVICES = 64;
struct DeviceTables {
    std::once_flag once, wide_once;
    int rc = 0, wide_rc = 0;
    u32* wide = nullptr;      // [WB_NT][WB_ROWS][WB_ROW_WORDS]: the wide comb's packed tables (2 MiB), made on first use
    u32* limbs = nullptr;     // [BASE_NT][30][128] signed comb tables 2^28 Ts .. Ts, [30][256]: the reference's table T, [30][SC_ROWS]: the lattice walk's comb
    u32* bytes = nullptr;     // [256][24]
};
DeviceTables g_tables[MAX_DEVICES];

int init_tables(DeviceTables& t)
{
    C25519_TRY(hipMalloc(&t.limbs, ALL_TBL_WORDS * sizeof(u32)));
    C25519_TRY(hipMalloc(&t.bytes, 256 * 24 * sizeof(u32)));
    k_gen_base_table<<<BASE_NT + (256 + SC_ROWS) / BASE_ROWS, BASE_ROWS, 0, nullptr>>>(t.limbs, t.bytes);
    C25519_TRY(hipGetLastError());
    C25519_TRY(hipStreamSynchronize(nullptr));
    return 0;
}

}  // namespace

namespace c25519_engine {

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

// the wide fixed-base comb of the current device (tunable BASE_COMB = 1), generated at its first use
int wide_tables(const u32** wide)
{
    int dev = 0;
    C25519_TRY(hipGetDevice(&dev));
    if (dev < 0 || dev >= MAX_DEVICES) return bad_arg("device ordinal out of range");
    DeviceTables& t = g_tables[dev];
    std::call_once(t.wide_once, [&] {
        t.wide_rc = [&]() -> int {
            C25519_TRY(hipMalloc(&t.wide, WB_TBL_WORDS * sizeof(u32)));
            k_gen_wide_table<<<WB_NT * WB_ROWS / 128, 128, 0, nullptr>>>(t.wide);
            C25519_TRY(hipGetLastError());
            C25519_TRY(hipStreamSynchronize(nullptr));
            return 0;
        }();
    });
    if (t.wide_rc) return t.wide_rc;
    *wide = t.wide;
    return 0;
}

}  // namespace c25519_engine

extern "C" {

int curve25519_dh_CalculatePublicKey_fast_dev(void* pk, void* sk, size_t n, void* stream_)
{
    C25519_API_CALL();
    if (!pk || !sk) return bad_arg("null pointer");
    if (int rc = check_dev_args(n, { pk, sk })) return rc;
    if (n == 0) return 0;
    hipStream_t stream = (hipStream_t)stream_;
    const u32* tbl = nullptr;
    C25519_RC(base_tables(&tbl, nullptr));
    const bool wide_comb = base_comb_wide();                  // every knob is read ONCE per call (another thread may turn it meanwhile)
    if (wide_comb && fixed_base_quad_for(n)) {                // four lanes per element
        const u32* wide = nullptr;
        C25519_RC(wide_tables(&wide));
        k_x25519_public_fast_quad<<<grid_for(n, quad::ELEMS_PER_WAVE), 64, 0, stream>>>(pk, sk, n, wide);
        C25519_TRY(hipGetLastError());
        return 0;
    }
    if (fixed_base_coop_for(n)) {                             // a few elements: one operation per wave
        if (wide_comb) {
            const u32* wide = nullptr;
            C25519_RC(wide_tables(&wide));
            k_x25519_public_fast_coop<true><<<(unsigned)n, 64, 0, stream>>>(pk, sk, n, wide, take_done_word(n));
        } else k_x25519_public_fast_coop<false><<<(unsigned)n, 64, 0, stream>>>(pk, sk, n, tbl, take_done_word(n));
        C25519_TRY(hipGetLastError());
        return 0;
    }
    void* w = nullptr;
    c25519_host::WorkLease lease;
    C25519_RC(lease.acquire(&w, proj_words(n) * sizeof(u32), stream));
    const ProjScratch scr = carve_proj((u32*)w, n);
    if (wide_comb) {
        const u32* wide = nullptr;
        C25519_RC(wide_tables(&wide));
        k_x25519_public_fast_mult<true><<<grid_for(n, WB_BLOCK), WB_BLOCK, 0, stream>>>(scr, sk, n, wide);
    } else {
        k_x25519_public_fast_mult<false><<<grid_for(n, bm_block_for(n)), bm_block_for(n), 0, stream>>>(scr, sk, n, tbl);
    }
    C25519_TRY(hipGetLastError());
    C25519_RC(launch_invert(scr, n, FinishX25519{ scr.a, pk, n }, stream));
    return lease.release();
}

}  // extern "C"
int c25519_engine::keypair_dev(void* pub, void* priv, const void* sk, const void* blinding, size_t n, hipStream_t stream)
{
    if (!pub || !priv || !sk) return bad_arg("null pointer");
    if (int rc = check_dev_args(n, { pub, priv, sk, blinding })) return rc;
    if (n == 0) return 0;
    const u32* tbl = nullptr;
    C25519_RC(base_tables(&tbl, nullptr));
    const bool wide_comb = base_comb_wide();                  // every knob is read ONCE per call (another thread may turn it meanwhile)
    if (!blinding && wide_comb && fixed_base_quad_for(n)) {   // four lanes per element
        const u32* wide = nullptr;
        C25519_RC(wide_tables(&wide));
        k_ed25519_keypair_quad<<<grid_for(n, quad::ELEMS_PER_WAVE), 64, 0, stream>>>(pub, priv, sk, n, wide);
        C25519_TRY(hipGetLastError());
        return 0;
    }
    if ((!blinding || wide_comb) && fixed_base_coop_for(n)) {   // a few elements: one operation per wave
        if (wide_comb) {
            const u32* wide = nullptr;
            C25519_RC(wide_tables(&wide));
            k_ed25519_keypair_coop<true><<<(unsigned)n, 64, 0, stream>>>(pub, priv, sk, n, wide, (const u32*)blinding, take_done_word(n),
                                                                         call_record_and_message(n, sk, 32, nullptr, 0));
        } else k_ed25519_keypair_coop<false><<<(unsigned)n, 64, 0, stream>>>(pub, priv, sk, n, tbl, nullptr, take_done_word(n),
                                                                               call_record_and_message(n, sk, 32, nullptr, 0));
        C25519_TRY(hipGetLastError());
        return 0;
    }
    void* w = nullptr;
    c25519_host::WorkLease lease;
    C25519_RC(lease.acquire(&w, proj_words(n) * sizeof(u32), stream));
    const ProjScratch scr = carve_proj((u32*)w, n);
    if (wide_comb) {
        const u32* wide = nullptr;
        C25519_RC(wide_tables(&wide));
        if (blinding) k_ed25519_keypair_mult<true, true><<<grid_for(n, WB_BLOCK), WB_BLOCK, 0, stream>>>(scr, priv, sk, n, wide, (const u32*)blinding);
        else k_ed25519_keypair_mult<false, true><<<grid_for(n, WB_BLOCK), WB_BLOCK, 0, stream>>>(scr, priv, sk, n, wide, nullptr);
    } else if (blinding)
        k_ed25519_keypair_mult<true, false><<<grid_for(n, bm_block_for(n)), bm_block_for(n), 0, stream>>>(scr, priv, sk, n, tbl, (const u32*)blinding);
    else
        k_ed25519_keypair_mult<false, false><<<grid_for(n, bm_block_for(n)), bm_block_for(n), 0, stream>>>(scr, priv, sk, n, tbl, nullptr);
    C25519_TRY(hipGetLastError());
    // pub[e] and priv[e][32..63] <- enc(A)
    C25519_RC(launch_invert(scr, n, FinishPack{ scr.a, scr.b, pub, n, 1, 0, priv, 2, 1 }, stream));
    return lease.release();
}

extern "C" {

int ed25519_CreateKeyPair_dev(void* pub, void* priv, const void* sk, size_t n, void* stream)
{
    C25519_API_CALL();
    return keypair_dev(pub, priv, sk, nullptr, n, (hipStream_t)stream);
}

int ed25519_CreateKeyPair_blinded_dev(void* pub, void* priv, const void* blinding, const void* sk, size_t n, void* stream)
{
    C25519_API_CALL();
    if (!blinding) return bad_arg("null blinding context");
    return keypair_dev(pub, priv, sk, blinding, n, (hipStream_t)stream);
}

}  // extern "C"
int c25519_engine::sign_dev(void* sig, const void* priv, const void* blinding, Msgs msgs, size_t n, hipStream_t stream)
{
    if (int rc = check_dev_args(n, { sig, priv, blinding })) return rc;
    if (n == 0) return 0;
    const u32* tbl = nullptr;
    C25519_RC(base_tables(&tbl, nullptr));
    const bool wide_comb = base_comb_wide();                  // every knob is read ONCE per call (another thread may turn it meanwhile)
    if (!blinding && wide_comb && fixed_base_quad_for(n)) {   // four lanes per element
        const u32* wide = nullptr;
        C25519_RC(wide_tables(&wide));
        k_ed25519_sign_quad<<<grid_for(n, quad::ELEMS_PER_WAVE), 64, 0, stream>>>(sig, priv, msgs, n, wide);
        C25519_TRY(hipGetLastError());
        return 0;
    }
    if ((!blinding || wide_comb) && fixed_base_coop_for(n)) {   // a few elements: one operation per wave
        const CallWords cw = msgs.offsets ? CallWords{} : call_record_and_message(n, priv, 64, msgs.base, msgs.fixed);
        if (wide_comb) {
            const u32* wide = nullptr;
            C25519_RC(wide_tables(&wide));
            k_ed25519_sign_coop<true><<<(unsigned)n, COOP_SHA_BLOCK, 0, stream>>>(sig, priv, msgs, n, wide, (const u32*)blinding, take_done_word(n), cw);
        } else k_ed25519_sign_coop<false><<<(unsigned)n, COOP_SHA_BLOCK, 0, stream>>>(sig, priv, msgs, n, tbl, nullptr, take_done_word(n), cw);
        C25519_TRY(hipGetLastError());
        return 0;
    }
    void* w = nullptr;
    const size_t sc_words = round_up(8 * n, 4);
    c25519_host::WorkLease lease;
    C25519_RC(lease.acquire(&w, (proj_words(n) + 2 * sc_words) * sizeof(u32), stream));
    const ProjScratch scr = carve_proj((u32*)w, n);
    u32* a_buf = (u32*)w + proj_words(n);
    u32* r_buf = a_buf + sc_words;
    if (wide_comb) {
        const u32* wide = nullptr;
        C25519_RC(wide_tables(&wide));
        if (blinding) k_ed25519_sign_mult<true, true><<<grid_for(n, WB_BLOCK), WB_BLOCK, 0, stream>>>(scr, a_buf, r_buf, priv, msgs, n, wide, (const u32*)blinding);
        else k_ed25519_sign_mult<false, true><<<grid_for(n, WB_BLOCK), WB_BLOCK, 0, stream>>>(scr, a_buf, r_buf, priv, msgs, n, wide, nullptr);
    } else if (blinding)
        k_ed25519_sign_mult<true, false><<<grid_for(n, bm_block_for(n)), bm_block_for(n), 0, stream>>>(scr, a_buf, r_buf, priv, msgs, n, tbl,
                                                                                      (const u32*)blinding);
    else
        k_ed25519_sign_mult<false, false><<<grid_for(n, bm_block_for(n)), bm_block_for(n), 0, stream>>>(scr, a_buf, r_buf, priv, msgs, n, tbl,
                                                                                       nullptr);
    C25519_TRY(hipGetLastError());
    // (the last two launches in one -- the shared inversion inside the workgroup, then h and S -- lost to this at every width:
    // profiles/r04_ab_sign_tail.txt)
    C25519_RC(launch_invert(scr, n, FinishPack{ scr.a, scr.b, sig, n, 2, 0, nullptr, 0, 0 }, stream));   // sig[e][0..31] = enc(R)
    k_ed25519_sign_finish<<<grid_for(n, ED_BLOCK), ED_BLOCK, 0, stream>>>(sig, priv, msgs, n, a_buf, r_buf);
    C25519_TRY(hipGetLastError());
    return lease.release();
}

extern "C" {

int ed25519_SignMessage_dev(void* sig, const void* priv, const void* msg, size_t msg_size, size_t n, void* stream)
{
    C25519_API_CALL();
    if (!sig || !priv || (!msg && msg_size)) return bad_arg("null pointer");
    return sign_dev(sig, priv, nullptr, Msgs{ (const uint8_t*)msg, msg_size, nullptr }, n, (hipStream_t)stream);
}

int ed25519_SignMessage_blinded_dev(void* sig, const void* priv, const void* blinding, const void* msg, size_t msg_size,
                                    size_t n, void* stream)
{
    C25519_API_CALL();
    if (!sig || !priv || !blinding || (!msg && msg_size)) return bad_arg("null pointer");
    return sign_dev(sig, priv, blinding, Msgs{ (const uint8_t*)msg, msg_size, nullptr }, n, (hipStream_t)stream);
}

int ed25519_SignMessage_ragged_dev(void* sig, const void* priv, const void* msgs, const uint64_t* offsets, size_t n,
                                   void* stream)
{
    C25519_API_CALL();
    if (!sig || !priv || !offsets) return bad_arg("null pointer");
    return sign_dev(sig, priv, nullptr, Msgs{ (const uint8_t*)msgs, 0, (const unsigned long long*)offsets }, n,
                    (hipStream_t)stream);
}

// one 192-byte blinding context from seed[0..seed_len) (device pointers)
int ed25519_Blinding_Init_dev(void* ctx, const void* seed, size_t seed_len, void* stream)
{
    C25519_API_CALL();
    if (!ctx || (!seed && seed_len)) return bad_arg("null pointer");
    if (int rc = check_dev_args(1, { ctx })) return rc;
    const u32* tbl = nullptr;
    C25519_RC(base_tables(&tbl, nullptr));
    if (base_comb_wide()) {
        const u32* wide = nullptr;
        C25519_RC(wide_tables(&wide));
        k_ed25519_blinding_init_coop<<<1, 64, 0, (hipStream_t)stream>>>((u32*)ctx, (const uint8_t*)seed, seed_len, wide, take_done_word(1));
    } else {
        k_ed25519_blinding_init<<<1, 256, 0, (hipStream_t)stream>>>((u32*)ctx, (const uint8_t*)seed, seed_len, tbl);
    }
    C25519_TRY(hipGetLastError());
    return 0;
}

}  // extern "C"
