// curve25519_amd/csrc/engine.hip -- gfx950 kernels and the C-ABI shim of the batched Curve25519 /
// Ed25519 engine.  One keypair / signature per lane; every arithmetic step of the path runs on the
// device.  Entry points are declared in include/curve25519_amd.h, include/curve25519_dh.h and
// include/ed25519_signature.h (each cites the reference prototype it replaces).
//
// The reference pays one field inversion (ecp_Inverse, 254 S + 11 M) per call (curve25519_dh.c:148,
// ed25519_sign.c:265); here it is shared between several elements with Montgomery's trick:
//   * X25519: a batch that fills the chip is two launches (k_x25519_ladder, then k_batch_invert<FinishX25519>); up to 2^16
//     elements it is ONE (k_x25519_fused: the workgroup's waves park their projective results in LDS and one wave inverts
//     them all); a call of a few elements runs one operation per WAVE (k_x25519_coop);
//   * Ed25519 operations are two or three launches on the caller's stream: a "mult" kernel leaves the
//     projective point in scratch, k_batch_invert (K elements per lane) writes the canonical bytes, and sign
//     adds a finish kernel that hashes enc(R) || pk || m and computes S.
//
// Build: curve25519_amd/build.py (hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -mllvm -pragma-unroll-threshold=131072 ...)
#include "capi_common.hpp"
#include "host_pipeline.hpp"
#include "lanes.cuh"
#include "verify_fast.cuh"
#include "coop25519.cuh"
#include "coop_ops.cuh"
#include "quad25519.cuh"

#include "../../include/curve25519_amd.h"
#include "../../include/curve25519_dh.h"
#include "../../include/ed25519_signature.h"

#include <algorithm>
#include <condition_variable>
#include <initializer_list>
#include <mutex>
#include <system_error>
#include <thread>
#include <vector>

using namespace c25519;

// per-call scratch, carved out of one slab (all sizes in u32 words per element)
constexpr size_t SCR_FE = 10;
struct ProjScratch {            // projective result + prefix products of the batched inversion
    u32 *a, *b, *z, *prefix;    // X25519 public_fast: a = numerator, z = denominator.  Edwards: a = X, b = Y, z = Z.
};

// ------------------------------------------------------------------------------------------------
// X25519   (curve25519_dh_CreateSharedKey / curve25519_dh_CalculatePublicKey)
// ------------------------------------------------------------------------------------------------
// Single launch: the eight waves of a workgroup finish their ladders, park (PX, PZ) in LDS, and wave 0 inverts all
// the workgroup's Z's with ONE exponentiation (eight elements per lane, Montgomery's trick, prefix products in LDS);
// then every lane finishes its own element.  The projective intermediates never leave the CU: HBM traffic is the
// API's 96 B/op plus the clamped-key write-back.   BASE9 (pk == nullptr): ladder on the base point u = 9.
#ifndef C25519_XF_BLOCK
#define C25519_XF_BLOCK 512
#endif
#ifndef C25519_XF_WAVES
#define C25519_XF_WAVES 4             // waves per SIMD the register allocator aims at (A/B: profiles/r02_ab_occupancy.txt)
#endif
constexpr int XF_BLOCK = C25519_XF_BLOCK;     // waves per workgroup = elements per inverting lane

// Opt-in measurement build (tools/cycle_probe.py; never the product): -DC25519_CYCLE_PROBE=1 makes every wave of
// k_x25519_fused stamp s_memtime (one tick = one shader cycle) at its phase boundaries -- entry, end of the ladder, behind
// the first barrier, behind the shared inversion, behind the second barrier, exit -- with the hardware slot it ran on,
// so that cycles per ladder step, the idle time of a workgroup's waves during the inversion and the clock of an
// UN-PROFILED run (kernel wall time / cycles) can be read; =2 additionally accumulates the ten sections of a ladder step.
#ifdef C25519_CYCLE_PROBE
constexpr int PROBE_WORDS = 20;
__device__ unsigned long long* g_cycle_probe = nullptr;
C25519_DEV unsigned long long probe_now()
{
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : : "memory");
    return t;
}
// the constant 100 MHz counter: (shader cycles) / (these ticks) * 100 MHz is the shader clock the wave ran at, with no
// host-side timing involved
C25519_DEV unsigned long long probe_realtime()
{
    unsigned long long t;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : : "memory");
    return t;
}
struct SectionTimer {
    unsigned long long *last, *acc;
    C25519_DEV void operator()(int id) const
    {
#if C25519_CYCLE_PROBE >= 2
        C25519_SCHED_FENCE();
        const unsigned long long t = probe_now();
        if (id >= 0) acc[id] += t - *last;
        *last = t;
        C25519_SCHED_FENCE();
#endif
    }
};
#define C25519_PROBE_STAMP(i) do { C25519_SCHED_FENCE(); probe_t[i] = probe_now(); C25519_SCHED_FENCE(); } while (0)
#else
#define C25519_PROBE_STAMP(i) do { } while (0)
#endif

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

// z <- 1 where z == 0 (mod p), returns all-ones in that case: a zero takes no part in a shared inversion and its
// "inverse" is forced to 0 afterwards, which is what the reference's z^(p-2) gives (curve25519_dh.c:148)
C25519_DEV u32 fe_zero_to_one(fe& z)
{
    u32 w[8], nz = 0;
    fe_to_words(w, z);
#pragma unroll
    for (int q = 0; q < 8; q++) nz |= w[q];
    const u32 is_zero = nz ? 0u : 0xffffffffu;
    fe one;
    fe_set_u32(one, 1);
    fe_select(z, is_zero, one, z);
    return is_zero;
}

// BLOCK lanes per workgroup = 64 x the elements per inverting lane.  XF_BLOCK (512) is the throughput shape; a batch that
// does not fill the chip with it runs narrower workgroups (x25519_block_for): 2^14 elements are 32 workgroups of 512 -- 32
// of 256 CUs, two waves per SIMD -- but 256 of 64, one wave on a SIMD of its own, which finishes in little more than half
// the time; the price, an inversion per 1 / 2 / 4 elements instead of 8, is 2-8 % more instructions.
template <bool BASE9, int BLOCK>
__global__ void __launch_bounds__(BLOCK, C25519_XF_WAVES) k_x25519_fused(void* out, const void* pk, void* sk, size_t n)
{
    constexpr int K = BLOCK / 64;            // elements per lane of the inverting wave
    __shared__ u32 zbuf[10 * BLOCK];      // PZ, later 1/PZ
    __shared__ u32 xbuf[10 * BLOCK];      // PX
    __shared__ u32 pbuf[(K > 1 ? K - 1 : 1) * 10 * 64];   // prefix products of the inverting wave
    const int tid = threadIdx.x;
    const size_t i = (size_t)blockIdx.x * BLOCK + tid;
    const bool active = i < n;
#ifdef C25519_CYCLE_PROBE
    unsigned long long probe_t[6] = {}, probe_sec[10] = {}, probe_last = 0;
    const unsigned long long probe_rt0 = probe_realtime();
#endif
    C25519_PROBE_STAMP(0);
    {
        fe PX, PZ;
        if (active) {
            u32 u[8] = { 9, 0, 0, 0, 0, 0, 0, 0 }, k[8];
            if (!BASE9) load32(u, pk, i);
            load32(k, sk, i);
            clamp_words(k);
            store32(sk, i, k);                   // the reference clamps in the caller's buffer
#ifdef C25519_CYCLE_PROBE
            x25519_ladder_xz<BASE9>(PX, PZ, u, k, SectionTimer{ &probe_last, probe_sec });
#else
            x25519_ladder_xz<BASE9>(PX, PZ, u, k);
#endif
        } else {
            fe_set_u32(PX, 0);
            fe_set_u32(PZ, 1);
        }
        C25519_PROBE_STAMP(1);
        lds_put_fe(zbuf, BLOCK, tid, PZ);
        lds_put_fe(xbuf, BLOCK, tid, PX);
    }
    __syncthreads();
    C25519_PROBE_STAMP(2);
    if (tid < 64) {
        fe acc, z, zero;
        fe_set_u32(zero, 0);
        u32 zero_mask = 0;
#pragma unroll 1
        for (int t = 0; t < K; t++) {
            lds_get_fe(z, zbuf, BLOCK, tid + 64 * t);
            zero_mask |= (fe_zero_to_one(z) & 1u) << t;
            if (t == 0) acc = z; else fe_mul(acc, acc, z);
            if (t < K - 1) lds_put_fe(pbuf + t * 640, 64, tid, acc);
        }
        fe inv;
        fe_invert(inv, acc);
#pragma unroll 1
        for (int t = K - 1; t >= 0; t--) {
            fe zi;
            const u32 was_zero = ((zero_mask >> t) & 1u) ? 0xffffffffu : 0u;
            if (t > 0) {
                fe p;
                lds_get_fe(p, pbuf + (t - 1) * 640, 64, tid);
                fe_mul(zi, inv, p);
                lds_get_fe(z, zbuf, BLOCK, tid + 64 * t);
                fe one;
                fe_set_u32(one, 1);
                fe_select(z, was_zero, one, z);
                fe_mul(inv, inv, z);
                fe_select(zi, was_zero, zero, zi);
            } else {
                fe_select(zi, was_zero, zero, inv);
            }
            lds_put_fe(zbuf, BLOCK, tid + 64 * t, zi);
        }
    }
    C25519_PROBE_STAMP(3);
    __syncthreads();
    C25519_PROBE_STAMP(4);
    if (active) {
        fe x, zi;
        u32 w[8];
        lds_get_fe(x, xbuf, BLOCK, tid);
        lds_get_fe(zi, zbuf, BLOCK, tid);
        fe_mul(x, x, zi);
        fe_to_words(w, x);
        store32(out, i, w);                      // written last: `out` may alias `pk`
    }
#ifdef C25519_CYCLE_PROBE
    C25519_PROBE_STAMP(5);
    if ((tid & 63) == 0 && g_cycle_probe) {
        unsigned long long* rec = g_cycle_probe + ((size_t)blockIdx.x * (BLOCK / 64) + tid / 64) * PROBE_WORDS;
        for (int q = 0; q < 6; q++) rec[q] = probe_t[q];
        // HW_ID (wave / SIMD / CU / SH / SE slot) and XCC_ID of the wave
        rec[6] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) | ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32);
        for (int q = 0; q < 10; q++) rec[7 + q] = probe_sec[q];
        rec[17] = probe_rt0;
        rec[18] = probe_realtime();
    }
#endif
}

// The ladder alone: (PX : PZ) to the struct-of-arrays scratch, for k_batch_invert<FinishX25519> behind it.  No LDS, no
// barrier: every wave is on its own, a finished wave's slot goes to the next workgroup at once.  (k_x25519_fused parks
// seven of a workgroup's eight waves at a barrier while wave 0 inverts -- and as every workgroup of a full launch takes
// the same time, both workgroups of a CU get there together: tools/cycle_probe.py, profiles/r04_cycle_probe.txt.)
constexpr int XL_BLOCK = 256;
template <bool BASE9>
__global__ void __launch_bounds__(XL_BLOCK, C25519_XF_WAVES) k_x25519_ladder(u32* X, u32* Z, const void* pk, void* sk, size_t n)
{
    const size_t i = (size_t)blockIdx.x * XL_BLOCK + threadIdx.x;
    if (i >= n) return;
#ifdef C25519_CYCLE_PROBE
    unsigned long long probe_t[6] = {};
    const unsigned long long probe_rt0 = probe_realtime();
#endif
    C25519_PROBE_STAMP(0);
    u32 u[8] = { 9, 0, 0, 0, 0, 0, 0, 0 }, k[8];
    if (!BASE9) load32(u, pk, i);
    load32(k, sk, i);
    clamp_words(k);
    store32(sk, i, k);                           // the reference clamps in the caller's buffer
    fe PX, PZ;
    x25519_ladder_xz<BASE9>(PX, PZ, u, k);
    C25519_PROBE_STAMP(1);
    soa_store_fe(X, n, i, PX);
    soa_store_fe(Z, n, i, PZ);
#ifdef C25519_CYCLE_PROBE
    C25519_PROBE_STAMP(5);
    if ((threadIdx.x & 63) == 0 && g_cycle_probe) {
        unsigned long long* rec = g_cycle_probe + (i / 64) * PROBE_WORDS;
        probe_t[2] = probe_t[3] = probe_t[4] = probe_t[1];
        for (int q = 0; q < 6; q++) rec[q] = probe_t[q];
        rec[6] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) | ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32);
        rec[17] = probe_rt0;
        rec[18] = probe_realtime();
    }
#endif
}

// The completion word of a call of ONE element through the host-pointer prototypes (capi_common.hpp: ThreadState::done_word): the
// call's last kernel stores `seq` into pinned host memory BEHIND its results -- by the thread that stored them, or behind a wave's
// own stores: the fence waits for every store of the wave -- and the calling thread, which spins on the word, returns 4.6 us before
// the runtime's event would let it (profiles/r06_launch_latency.txt).  word == nullptr: nobody is waiting that way.
struct DoneWord { u32* word; u32 seq; };
C25519_DEV void signal_done(const DoneWord& d)
{
    if (d.word) {
        __threadfence_system();
        __hip_atomic_store(d.word, d.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// One operation per WAVE (coop25519.cuh): what a call of a few elements runs -- the reference's own single-call
// prototypes above all.  Ladder, doublings, inversion and the last multiplication are cooperative (a field element
// limb-per-lane, up to four products at a time); only the decoding of the inputs and the canonical encoding of the result
// are the batch kernels' per-lane code, run by every lane on the same values.
template <bool BASE9>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 4))) k_x25519_coop(void* out, const void* pk, void* sk, size_t n, DoneWord done, CallWords cw)
{
    __shared__ __attribute__((aligned(16))) u32 lds[coop::ROWQ_OFF];
    if (blockIdx.x >= n) return;
    coop::x25519_one<BASE9>(lds, coop::make_lane(threadIdx.x), out, pk, sk, blockIdx.x, &cw);
    if (threadIdx.x == 0) signal_done(done);
}

// ... and on TWO waves per element (coop::x25519_two_waves: a ladder step in two product levels -- the differential addition with
// x1 times the sum carried along on one wave, the doubling on the other, one workgroup barrier per step): what ONE
// curve25519_dh_CreateSharedKey call and calls of up to 512 run -- 183 -> 168 us per call
__global__ void __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(1, 4))) k_x25519_coop2(void* out, const void* pk, void* sk, size_t n, DoneWord done, CallWords cw)
{
    __shared__ __attribute__((aligned(16))) u32 lds[coop::X2_LDS_WORDS];
    if (blockIdx.x >= n) return;
    coop::x25519_two_waves(lds, out, pk, sk, blockIdx.x, &cw);
    if (threadIdx.x == 0) signal_done(done);               // (wave 0 stores; wave 1 has left inside)
}

// FOUR LANES per element (quad25519.cuh): what a call of 2^12 .. 2^14 elements runs -- too many for a wave each, too few to
// give every SIMD a wave of one-lane elements (2^14 elements are 256 such waves on 1024 SIMDs).  A quad runs one product of a
// ladder step per lane and level, operands exchanged with v_mov_b32_dpp quad_perm; 16 elements per wave, one wave per
// workgroup, inversion and encoding in the same launch: no LDS, no scratch, no barrier.
template <bool BASE9>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 2))) k_x25519_quad(void* out, const void* pk, void* sk, size_t n)
{
    const size_t e = (size_t)blockIdx.x * quad::ELEMS_PER_WAVE + (threadIdx.x >> 2);
    if (e >= n) return;                                       // (whole quads leave: the exchanges stay inside a quad)
    quad::x25519_element<BASE9>(out, pk, sk, e);
}

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
#ifndef C25519_ED_BLOCK
#define C25519_ED_BLOCK 256
#endif
#ifndef C25519_VI_WAVES
#define C25519_VI_WAVES 2            // waves per SIMD the register allocator aims at: Verify_Init ...
#endif
#ifndef C25519_VC_WAVES
#define C25519_VC_WAVES 2            // ... and Verify_Check (A/B: profiles/r02_ab_occupancy.txt)
#endif
constexpr int ED_BLOCK = C25519_ED_BLOCK;
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

C25519_DEV void store_proj(const ProjScratch& scr, size_t n, size_t i, const ge_ext& S)
{
    soa_store_fe(scr.a, n, i, S.X);
    soa_store_fe(scr.b, n, i, S.Y);
    soa_store_fe(scr.z, n, i, S.Z);
}

// The fixed-base kernels come in two shapes (tunable BASE_COMB, A/B: profiles/r05_ab_base_comb.txt):
//   WIDE = false  the 8 x 32 signed comb, eight tables staged in 120 KiB of LDS per 1024-lane workgroup: 31 additions + 3 doublings;
//   WIDE = true   the 13 x 20 signed comb of ge25519.cuh read through L2: 19 additions + 4 doublings, 256-lane workgroups, the
//                 only LDS the lanes' parked column numbers (10 KiB).
constexpr int WB_BLOCK = 256;
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
template <bool WIDE>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 4)))
k_ed25519_keypair_coop(void* pub, void* priv, const void* sk, size_t n, const u32* __restrict__ g_tbl,
                       const u32* __restrict__ blind_ctx, DoneWord done)
{
    __shared__ __attribute__((aligned(16))) u32 lds[coop::LDS_WORDS];
    if (blockIdx.x >= n) return;
    coop::keypair_one<WIDE>(lds, coop::make_lane(threadIdx.x), pub, priv, sk, blockIdx.x, g_tbl, blind_ctx);
    if (threadIdx.x == 0) signal_done(done);
}

// curve25519_dh_CalculatePublicKey_fast (curve25519_dh.c:162-189): S = clamp(sk) * B on the Edwards side, u = (Z + Y) / (Z - Y)
template <bool WIDE>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 4)))
k_x25519_public_fast_coop(void* pk, void* sk, size_t n, const u32* __restrict__ g_tbl, DoneWord done)
{
    __shared__ __attribute__((aligned(16))) u32 lds[coop::LDS_WORDS];
    if (blockIdx.x >= n) return;
    coop::public_fast_one<WIDE>(lds, coop::make_lane(threadIdx.x), pk, sk, blockIdx.x, g_tbl);
    if (threadIdx.x == 0) signal_done(done);
}

template <bool WIDE>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 4)))
k_ed25519_sign_coop(void* sig, const void* priv, Msgs msgs, size_t n, const u32* __restrict__ g_tbl,
                    const u32* __restrict__ blind_ctx, DoneWord done)
{
    __shared__ __attribute__((aligned(16))) u32 lds[coop::LDS_WORDS];
    if (blockIdx.x >= n) return;
    coop::sign_one<WIDE>(lds, coop::make_lane(threadIdx.x), sig, priv, msgs, blockIdx.x, g_tbl, blind_ctx);
    if (threadIdx.x == 0) signal_done(done);
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
    coop::blinding_init_one(lds, coop::make_lane(threadIdx.x), ctx, seed, seed_len, wide);
    if (threadIdx.x == 0) signal_done(done);
}

// ed25519_Verify_Init (ed25519_verify.c:179-232): decompress -A (inverted parity :192-195, no validation) and
// fill the key's 16-row 4-fold table.  `tables` holds n tables of Tbl's format, `stride_words` apart.
template <typename Tbl>
__global__ void __launch_bounds__(ED_BLOCK, C25519_VI_WAVES) k_ed25519_verify_init(const void* pk, size_t n, u32* tables,
                                                                      size_t stride_words)
{
    const size_t i = (size_t)blockIdx.x * ED_BLOCK + threadIdx.x;
    if (i >= n) return;
    u32 pkw[8];
    load32(pkw, pk, i);
    ge_ext Q;
    ed_decode_neg_key(Q, pkw);
    const Tbl tbl{ tables + i * stride_words };
    qtable_build(tbl, Q);
}

// ed25519_Verify_Check (ed25519_verify.c:287-313), first part: h = H(enc(R) || pk || m) mod L canonical;
// s = raw 256 bits (no s < L check, :308); T = s*B + h*(-A) projective.  The comparison with enc(R) happens in
// k_batch_invert<FinishVerify>.
template <typename Tbl>
C25519_DEV void verify_check_lane(const ProjScratch& scr, size_t n, size_t i, const void* sig, const u32 (&pkw)[8],
                                  const Msgs& msgs, const Tbl& tbl, const u32* lds_tbl)
{
    u32 Sw[8], h[8], Rw[8];
    load32(Rw, sig, 2 * i);
    ed_hram(h, Rw, pkw, msgs.ptr(i), msgs.len(i));
    sc_mod(h);
    load32(Sw, sig, 2 * i + 1);
    ge_ext T;
    ge_poly_mult(T, Sw, h, tbl, lds_tbl);
    store_proj(scr, n, i, T);
}

template <typename Tbl>
__global__ void __launch_bounds__(ED_BLOCK, C25519_VC_WAVES) k_ed25519_verify_check(ProjScratch scr, const void* sig, const void* pk,
                                                                       Msgs msgs, size_t n,
                                                                       const u32* __restrict__ g_tbl, u32* tables,
                                                                       size_t stride_words)
{
    __shared__ __attribute__((aligned(16))) u32 lds_tbl[PA_WORDS * 256];
    lds_stage_words(lds_tbl, g_tbl + REF_TBL_OFFSET, REF_TBL_WORDS);
    const size_t i = (size_t)blockIdx.x * ED_BLOCK + threadIdx.x;
    if (i >= n) return;
    u32 pkw[8];
    load32(pkw, pk, i);
    const Tbl tbl{ tables + i * stride_words };
    verify_check_lane(scr, n, i, sig, pkw, msgs, tbl, lds_tbl);
}

// ---- the lattice fast path (verify_fast.cuh) ---------------------------------------------------------------------------
// Four kernels.  scalars -> points -> walk decide every element whose key is on the curve (and whose short
// vector fits the walk: a random one practically always does); the elements they cannot decide are collected in a list
// and k_ed25519_verify_slow runs the reference's own operation order for exactly those.
// Per-element hand-over, struct-of-arrays: sigma_cols[SIGMA_WORDS] (sigma's signed comb columns), rho[5], tau[5] (biased), a flag word
//   bit 0  R decodes canonically onto the curve      bit 1  the key is on the curve
//   bit 2  the short vector fits the walk             bit 3  tau < 0
//   bit 4  the element is on the slow list            bits 8..13  top nonzero digit of the element's scalars
// (FastScratch, the scratch of the lattice path, and the FLAG_* bits: coop_ops.cuh)
constexpr int FS_BLOCK = 256;
#ifndef C25519_VW_WAVES
#define C25519_VW_WAVES 2            // waves per SIMD the register allocator aims at: the walk kernel (rows prefetched) ...
#endif
#ifndef C25519_WALK_BLOCK
#define C25519_WALK_BLOCK C25519_ED_BLOCK       // lanes per walk workgroup (they share one staged comb table)
#endif
constexpr int WALK_BLOCK = C25519_WALK_BLOCK;
#ifndef C25519_WALK_SORTED
#define C25519_WALK_SORTED 1         // A/B switch: 0 = the walk's lane j takes element j
#endif
#ifndef C25519_VD_WAVES
#define C25519_VD_WAVES 3            // ... and the point decoding + table kernel
#endif

C25519_DEV void verify_scalars_lane(const FastScratch& fs, const void* sig, const void* pk, const Msgs& msgs, size_t n, size_t i)
{
    u32 pkw[8], Rw[8], Sw[8], cols[SIGMA_WORDS], rho[5], tau[5], tau_neg;
    load32(pkw, pk, i);
    load32(Rw, sig, 2 * i);
    load32(Sw, sig, 2 * i + 1);
    const u32 lat_ok = ed_verify_fast_scalars(cols, rho, tau, tau_neg, pkw, Rw, Sw, msgs.ptr(i), msgs.len(i), fs.lat_cap_bits);
#pragma unroll
    for (int w = 0; w < SIGMA_WORDS; w++) fs.sigma[(size_t)w * n + i] = cols[w];
#pragma unroll
    for (int w = 0; w < 5; w++) { fs.rho[(size_t)w * n + i] = rho[w]; fs.tau[(size_t)w * n + i] = tau[w]; }
    const int top = lat_ok ? walk_top_digit(tau, rho) : 0;
    fs.flags[i] = (lat_ok & FLAG_FITS) | (tau_neg & FLAG_TAU_NEG) | ((u32)top << 8);
}


// step 1: hash, short lattice vector, sigma -- integer work only
__global__ void __launch_bounds__(FS_BLOCK) k_ed25519_verify_fast_scalars(FastScratch fs, const void* sig, const void* pk,
                                                                          Msgs msgs, size_t n)
{
    const size_t i = (size_t)blockIdx.x * FS_BLOCK + threadIdx.x;
    if (i == 0) fs.slow_count[0] = fs.slow_count[1] = fs.slow_count[2] = 0;
    if (i >= n) return;
    verify_scalars_lane(fs, sig, pk, msgs, n, i);
}

// step 2: the two points of an element, one per lane: lane j < n decodes key j, lane n + j decodes R of signature j (a
// square root each), then builds that point's window table.  2n lanes, 168 registers: three waves per SIMD, no spills.
__global__ void __launch_bounds__(ED_BLOCK, C25519_VD_WAVES) k_ed25519_verify_fast_points(FastScratch fs, const void* sig, const void* pk,
                                                                                          size_t n)
{
    const size_t j = (size_t)blockIdx.x * ED_BLOCK + threadIdx.x;
    if (j >= 2 * n) return;
    const bool is_r = j >= n;
    const size_t e = is_r ? j - n : j;
    u32 w[8];
    if (is_r) load32(w, sig, 2 * e); else load32(w, pk, e);
    const u32 f = fs.flags[e];
    const u32 tau_neg = (f & FLAG_TAU_NEG) ? 0xffffffffu : 0u;
    fe X, Y;
    const u32 ok = ed_verify_fast_decode(X, Y, w, is_r ? 0xffffffffu : 0u, tau_neg);
#if C25519_WALK_SORTED
    if (!is_r) {                                                   // the walk's order (see FastScratch::order)
        const bool is_long = ((f >> 8) & 63u) > 32u;
        const u32 pos = is_long ? (u32)n - 1u - atomicAdd(fs.slow_count + 2, 1u) : atomicAdd(fs.slow_count + 1, 1u);
        fs.order[pos] = (u32)e;
    }
#endif
    if (is_r) {
        if (ok) atomicOr(&fs.flags[e], FLAG_R_OK);
    } else if (ok && (f & FLAG_FITS)) {
        atomicOr(&fs.flags[e], FLAG_KEY_OK);
    } else {                                                      // an element the walk cannot decide: on the slow list
        atomicOr(&fs.flags[e], ok ? FLAG_KEY_OK | FLAG_SLOW : FLAG_SLOW);
        fs.slow_list[atomicAdd(fs.slow_count, 1u)] = (u32)e;      // (the compiler aggregates this per wave)
    }
    // the point's window table, right here: a table is 1152 bytes of 16-byte stores scattered over as many cache lines, and
    // they hide under the other waves' square roots (measured with the earlier 160-byte rows: in a kernel of their own 1.3 ms
    // with the SIMDs idle half the time; in front of the walk, inside its kernel, 1.0 ms; here 0.6 ms --
    // profiles/r03_ab_verify_structure.txt).
    // (An element that turns out to be on the slow list gets tables nobody reads: the key lane cannot tell the R lane in time.)
    wtable_build(fs.tables + e * FAST_TABLE_WORDS + (is_r ? WTABLE_WORDS : 0), X, Y);
}

// step 3: the walk and the neutral-element test (ge_walk_is_neutral).  Beside the accumulator point only the round's two
// packed table rows live in registers -- fetched at the top of the round, unpacked field by field when the additions want
// them -- ; the scalars are fetched a word at a time, LDS rows a field at a time: 216 registers, two waves per SIMD, no
// spills.  The kernel is VALU-bound: a SIMD has a VALU instruction executing in 97 % of the shader's cycles
// (SQ_ACTIVE_INST_VALU * 4 / 1024 against GRBM_GUI_ACTIVE / 8, profiles/r03_pmc.txt), and it measured the same at two,
// three (154 registers without the prefetch) and four waves per SIMD.
__global__ void __launch_bounds__(WALK_BLOCK, C25519_VW_WAVES) k_ed25519_verify_fast_walk(FastScratch fs, int* verdict, size_t n,
                                                                                        const u32* __restrict__ g_tbl)
{
    __shared__ __attribute__((aligned(16))) u32 lds_tbl[SC_TBL_WORDS];
    lds_stage_words(lds_tbl, g_tbl + SC_TBL_OFFSET, SC_TBL_WORDS);
    const size_t lane = (size_t)blockIdx.x * WALK_BLOCK + threadIdx.x;
#if C25519_WALK_SORTED
    const size_t i = lane < n ? fs.order[lane] : n;
#else
    const size_t i = lane;
#endif
    const u32 f = i < n ? fs.flags[i] : FLAG_SLOW;
    const bool walks = !(f & FLAG_SLOW);
    // the wave walks from its longest element's first digit (the others' digits above their own are zero)
    int top = walks ? (int)((f >> 8) & 63u) : 0;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const int other = __shfl_xor(top, o);
        top = other > top ? other : top;
    }
    top = __builtin_amdgcn_readfirstlane(top);           // wave-uniform by construction: let the walk's loops be scalar ones
    if (!walks) return;
#ifdef C25519_WALK_TABLE_ALIAS                           // TIMING ONLY (wrong verdicts): every element reads one of 1024 tables, L2-resident
    const u32* tq = fs.tables + (i & 1023) * FAST_TABLE_WORDS;
#else
    const u32* tq = fs.tables + i * FAST_TABLE_WORDS;
#endif
    const WalkScalars sc{ fs.sigma, fs.tau, fs.rho, n, i };
    const u32 neutral = ge_walk_is_neutral(sc, tq, tq + WTABLE_WORDS, lds_tbl, top < 8 ? 8 : top);
    verdict[i] = (neutral & f & FLAG_R_OK) ? 1 : 0;
}

// Batches of 2^11 .. 2^15 signatures leave most of the chip idle under one-lane kernels (2^14 elements: 256 waves on 1024 SIMDs), so
// their path is shaped for the LENGTH of the chain, not for instructions per element:
//  * k_ed25519_verify_quad_prep -- ONE launch for steps 1 and 2: the first workgroups hash and reduce (50 us), the others decode the
//    two points of every element and build their window tables (92 us) AT THE SAME TIME.  The points cannot know tau's sign yet:
//    they tabulate the key as decoded, and the walk flips the rows' signs where tau < 0.  Each lane reports its point in a word of
//    its own (pflags), so nothing here is ordered against the scalar workgroups.
//  * k_ed25519_verify_quad_walk -- step 3 on QUADS (quad25519.cuh: quad::walk_is_neutral): four lanes per element walk an addition
//    in two product levels and a doubling in a level of squarings and one of products (~2.3 x shorter than a lane's); it also
//    makes the slow list (an element the walk cannot decide: off-curve key, over-long vector) for step 5 behind it.  64 elements
//    (four waves) per workgroup share one staged comb table; element order (no long / short sorting: 16 elements per wave).
__global__ void __launch_bounds__(ED_BLOCK, C25519_VD_WAVES) k_ed25519_verify_quad_prep(FastScratch fs, const void* sig, const void* pk,
                                                                                         Msgs msgs, size_t n, unsigned scalar_blocks)
{
    static_assert(FS_BLOCK == ED_BLOCK, "one workgroup shape for both roles");
    if (blockIdx.x < scalar_blocks) {
        const size_t i = (size_t)blockIdx.x * FS_BLOCK + threadIdx.x;
        if (i == 0) fs.slow_count[0] = fs.slow_count[1] = fs.slow_count[2] = 0;
        if (i >= n) return;
        verify_scalars_lane(fs, sig, pk, msgs, n, i);
        return;
    }
    const size_t j = (size_t)(blockIdx.x - scalar_blocks) * ED_BLOCK + threadIdx.x;
    if (j >= 2 * n) return;
    const bool is_r = j >= n;
    const size_t e = is_r ? j - n : j;
    u32 w[8];
    if (is_r) load32(w, sig, 2 * e); else load32(w, pk, e);
    fe X, Y;
    const u32 ok = ed_verify_fast_decode(X, Y, w, is_r ? 0xffffffffu : 0u, 0u);
    fs.pflags[j] = ok;
    wtable_build(fs.tables + e * FAST_TABLE_WORDS + (is_r ? WTABLE_WORDS : 0), X, Y);
}

constexpr int QW_BLOCK = 256;
__global__ void __launch_bounds__(QW_BLOCK) __attribute__((amdgpu_waves_per_eu(1, 2)))
k_ed25519_verify_quad_walk(FastScratch fs, int* verdict, size_t n, const u32* __restrict__ g_tbl)
{
    __shared__ __attribute__((aligned(16))) u32 lds_tbl[SC_TBL_WORDS];
    lds_stage_words(lds_tbl, g_tbl + SC_TBL_OFFSET, SC_TBL_WORDS);
    const size_t i = (size_t)blockIdx.x * (QW_BLOCK / 4) + (threadIdx.x >> 2);
    const u32 f = i < n ? fs.flags[i] : 0u;
    const u32 key_ok = i < n ? fs.pflags[i] : 0u, r_ok = i < n ? fs.pflags[n + i] : 0u;
    const bool walks = (f & FLAG_FITS) && key_ok;
    int top = walks ? (int)((f >> 8) & 63u) : 0;           // the wave walks from its longest element's first digit
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const int other = __shfl_xor(top, o);
        top = other > top ? other : top;
    }
    top = __builtin_amdgcn_readfirstlane(top);
    const quad::Roles R = quad::roles();
    if (!walks) {                                         // (whole quads leave)
        if (i < n && R.is0) fs.slow_list[atomicAdd(fs.slow_count, 1u)] = (u32)i;
        return;
    }
    const u32* tq = fs.tables + i * FAST_TABLE_WORDS;
    const WalkScalars sc{ fs.sigma, fs.tau, fs.rho, n, i };
    const u32 q_flip = (f & FLAG_TAU_NEG) ? 0xffffffffu : 0u;
    const u32 neutral = quad::walk_is_neutral(sc, tq, tq + WTABLE_WORDS, lds_tbl, top < 8 ? 8 : top, R, q_flip);
    if (R.is0) verdict[i] = (neutral & r_ok) ? 1 : 0;
}

// The whole lattice path of ONE element in ONE launch, for a call of a few elements: a workgroup of THREE waves per element
// (coop::verify_three_waves, coop_ops.cuh: wave 0 hashes and reduces while wave 1 takes the two square roots; then the three
// products of sigma*B + tau*Q + rho*(-R) = O side by side, a wave each; wave 0 adds and tests).  Three launches ran
// 40 + 83 + 91 us one after the other for one signature; this is ~60 + ~60.  The host zeroes the slow list's counter in front
// of the launch.
__global__ void __launch_bounds__(192) __attribute__((amdgpu_waves_per_eu(1, 2)))
k_ed25519_verify_one_per_group(FastScratch fs, int* verdict, const void* sig, const void* pk, Msgs msgs, size_t n,
                               const u32* __restrict__ g_tbl)
{
    __shared__ __attribute__((aligned(16))) u32 lds_all[coop::V3_LDS_WORDS];
    __shared__ u32 park[40], hand[4];
    if (blockIdx.x >= n) return;
    coop::verify_three_waves(lds_all, park, hand, fs, verdict, sig, pk, msgs, n, blockIdx.x, g_tbl);
}

// step 5: the elements on the slow list (off-curve keys -- the reference does not reject them, so neither may we -- and
// the practically nonexistent over-long vectors), one per lane, in the reference's order (ed_verify_reference_order),
// behind the walk on the same stream.  The grid covers the worst case (every element listed); workgroups beyond the
// list's end read the counter and leave: with honest keys that is all of them and costs ~10 us.  A batch with garbage keys
// in it pays one reference-order verification's latency (~1.3 ms) on top.
// (Tried and dropped: the kernel on a second, high-priority stream beside the walk -- its workgroups only ever found room
// when the walk's last round drained, profiles/r03_ab_verify_structure.txt; a fixed small grid striding over the list --
// any loop around the body makes the compiler keep ~60 field constants in registers across trips: 268 instead of 200.)
__global__ void __launch_bounds__(ED_BLOCK, 2) k_ed25519_verify_slow(FastScratch fs, int* verdict, const void* sig, const void* pk,
                                                                     Msgs msgs, const u32* __restrict__ g_tbl, DoneWord done)
{
    // done: a call of ONE element only (its list holds at most that element, which thread 0 of block 0 then decides)
    const u32 count = *fs.slow_count;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        *fs.slow_report = count;
        if (count == 0) signal_done(done);
    }
    if ((size_t)blockIdx.x * ED_BLOCK >= count) return;
    __shared__ __attribute__((aligned(16))) u32 lds_tbl[PA_WORDS * 256];
    lds_stage_words(lds_tbl, g_tbl + REF_TBL_OFFSET, REF_TBL_WORDS);
    const size_t k = (size_t)blockIdx.x * ED_BLOCK + threadIdx.x;
    if (k >= count) return;
    const size_t i = fs.slow_list[k];
    u32 pkw[8], Rw[8], Sw[8];
    load32(pkw, pk, i);
    load32(Rw, sig, 2 * i);
    load32(Sw, sig, 2 * i + 1);
    verdict[i] = ed_verify_reference_order(pkw, Rw, Sw, msgs.ptr(i), msgs.len(i), fs.tables + i * FAST_TABLE_WORDS, lds_tbl);
    if (k == 0) signal_done(done);
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
                                                                              size_t n, const u32* __restrict__ g_tbl,
                                                                              const u32* __restrict__ wide_ok)
{
    if (wide_ok && *wide_ok) return;                       // k_ed25519_verify_check_wide decides this batch
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
    lds_stage_words(lds_tbl, g_tbl + REF_TBL_OFFSET, REF_TBL_WORDS);   // ends with __syncthreads()
    const size_t i = (size_t)blockIdx.x * ED_BLOCK + threadIdx.x;
    if (i >= n) return;
    u32 pkw[8];
#pragma unroll
    for (int j = 0; j < 8; j++) pkw[j] = ctx[j];
    const QTableLds tbl{ lds_q };
    verify_check_lane(scr, n, i, sig, pkw, msgs, tbl, lds_tbl);
}

// ed25519_Verify_Init for a call of a few keys: one key per wave.  The square root by every lane on the same value (one lane's
// code: a cooperative one would be no faster), the table by the whole wave (coop::qtable_build_coop).  501 us per call in the
// per-lane kernel (a lone lane's 192 doublings), ~130 here.
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 4)))
k_ed25519_verify_init_coop(const void* pk, size_t n, u32* ctx_rows /* n contexts, stride_words apart, the 16 rows of each */, size_t stride_words,
                           DoneWord done)
{
    __shared__ __attribute__((aligned(16))) u32 lds[coop::Q_LDS_WORDS];
    if (blockIdx.x >= n) return;
    u32* rows = ctx_rows + blockIdx.x * stride_words;
    coop::verify_init_one(lds, coop::make_lane(threadIdx.x), pk, blockIdx.x, rows);
    if (threadIdx.x < 8) rows[(int)threadIdx.x - 8] = ((const u32*)pk)[blockIdx.x * 8 + threadIdx.x];   // the context's first 32 bytes: the key
    if (threadIdx.x == 0) signal_done(done);               // (rows and key are this one wave's stores: the fence waits for them all)
}

// ed25519_Verify_Check for a call of a few pairs (the reference's prototype is a call of ONE): one pair per wave, the
// reference's own operation order (coop::poly_mult), one shared-nothing inversion per pair.  454 us per call in the per-lane
// kernel above (a lone lane walks 63 doublings and 96 additions); ~125 here.
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 4)))
k_ed25519_verify_check_coop(int* verdict, const void* sig, const u32* __restrict__ ctx, Msgs msgs, size_t n, const u32* __restrict__ g_tbl,
                            DoneWord done)
{
    __shared__ __attribute__((aligned(16))) u32 lds[coop::Q_LDS_WORDS];
    if (blockIdx.x >= n) return;
    coop::verify_check_one(lds, coop::make_lane(threadIdx.x), verdict, sig, ctx, msgs, blockIdx.x, g_tbl + REF_TBL_OFFSET);
    if (threadIdx.x == 0) signal_done(done);
}

// ---- one key, a big batch: both scalars over wide combs ------------------------------------------------------------------
// With ONE key for the whole batch the double-scalar product T = s*B + h*(-A) is two FIXED-base products: the base point's
// wide comb (ge25519.cuh) and one built for -A the same way, walked together -- 39 additions and 4 doublings per signature
// instead of the reference order's 255 doublings and 95 additions (ed25519_verify.c:243-280).  For a key ON the curve any
// evaluation of the group law gives the same point T, hence the same enc(T) and the same verdict; so this path decides
// a batch only when (a) the context is byte for byte what Verify_Init computes for its key bytes (a context is caller
// storage: one that was written by anything else keeps the kernel above, which reads its rows as they are, like the
// reference) and (b) the key decompresses onto the curve.  k_ed25519_verify_ctx_prepare establishes both in block 0 -- one
// lane rebuilds the 16 rows, as Verify_Init did -- while the other blocks generate the key's comb rows (the work of
// k_gen_wide_table, 0.6 ms); worth it from 2^16 signatures per call (tunable ONE_KEY_WIDE).
// `remembered` (KEEP_CTX_WORDS + 1 words behind the key's comb, in a buffer that outlives the call): the context the comb was
// built for and a state word -- 0 nothing yet, 1 remembered but not eligible, 2 remembered and eligible.  The reference's use is
// ONE Verify_Init and MANY Verify_Check calls (ed25519_verify.c:282-286): a call whose context equals the remembered bytes skips
// all of the preparation (every block finds that out for itself: 2080 bytes out of L2); k_ed25519_verify_ctx_remember, behind
// this kernel on the stream, writes the bytes down.
constexpr int KEEP_CTX_WORDS = 2080 / 4;
// build_if_new = 0 (one block): only ask whether the context is the remembered one -- what calls below the ONE_KEY_WIDE size do:
// a remembered comb costs them nothing, a new one would cost more than they take.
__global__ void __launch_bounds__(128) k_ed25519_verify_ctx_prepare(u32* wide_key /*[WB_NT][WB_ROWS][WB_ROW_WORDS]*/, u32* check_rows /*[16][32]*/,
                                                                     u32* wide_ok, const u32* __restrict__ ctx,
                                                                     const u32* __restrict__ remembered, int build_if_new)
{
    {
        int same = remembered[KEEP_CTX_WORDS] != 0;
        for (int w = threadIdx.x; w < KEEP_CTX_WORDS; w += 128) same = same && remembered[w] == ctx[w];
        if (__syncthreads_and(same)) {
            if (blockIdx.x == 0 && threadIdx.x == 0) *wide_ok = remembered[KEEP_CTX_WORDS] == 2 ? 1u : 0u;
            return;
        }
        if (!build_if_new) {
            if (blockIdx.x == 0 && threadIdx.x == 0) *wide_ok = 0u;
            return;
        }
    }
    if (blockIdx.x == 0) {
        if (threadIdx.x != 0) return;
        u32 pkw[8];
#pragma unroll
        for (int j = 0; j < 8; j++) pkw[j] = ctx[j];
        ge_ext Q;
        u32 yw[8];
#pragma unroll
        for (int i = 0; i < 8; i++) yw[i] = pkw[i];
        const u32 parity = yw[7] >> 31;
        yw[7] &= 0x7fffffffu;
        fe_from_words(Q.Y, yw);
        const u32 on_curve = ge_calc_x_checked(Q.X, Q.Y, ~parity);     // ed_decode_neg_key, keeping the square root's verdict
        fe_mul(Q.T, Q.X, Q.Y);
        fe_set_u32(Q.Z, 1);
        qtable_build(QTableCanon{ check_rows }, Q);
        u32 diff = 0;
        for (int w = 0; w < 16 * 32; w++) diff |= check_rows[w] ^ ctx[8 + w];
        *wide_ok = (on_curve && diff == 0) ? 1u : 0u;
        return;
    }
    const u32 g = (blockIdx.x - 1) * 128 + threadIdx.x;       // table * WB_ROWS + row
    const int table = (int)(g / WB_ROWS);
    // -A in affine precomputed form = row 1 of the context (Verify_Init stores the decompressed key with Z = 1); if the context
    // is not Verify_Init's, block 0 says so and nobody reads these rows
    ge_pa P;
    {
        u32 w[8];
#pragma unroll
        for (int f = 0; f < 3; f++) {
#pragma unroll
            for (int j = 0; j < 8; j++) w[j] = ctx[8 + 32 + 8 * f + j];
            fe_from_words(f == 0 ? P.ypx : f == 1 ? P.ymx : P.t2d, w);
        }
    }
    u32 rows[3][8];
    ge_signed_comb_row_of(rows, P, g % WB_ROWS, (WB_NT - 1 - table) * WB_STEP, WB_TEETH, WB_COLS);
    uint4* out = reinterpret_cast<uint4*>(wide_key + (size_t)g * WB_ROW_WORDS);
#pragma unroll
    for (int f = 0; f < 3; f++) {
        out[2 * f] = make_uint4(rows[f][0], rows[f][1], rows[f][2], rows[f][3]);
        out[2 * f + 1] = make_uint4(rows[f][4], rows[f][5], rows[f][6], rows[f][7]);
    }
    out[6] = make_uint4(2, 0, 0, 0);                          // 2Z, as in k_gen_wide_table
    out[7] = make_uint4(0, 0, 0, 0);
}

__global__ void __launch_bounds__(128) k_ed25519_verify_ctx_remember(u32* remembered, const u32* __restrict__ ctx, const u32* __restrict__ wide_ok)
{
    for (int w = threadIdx.x; w < KEEP_CTX_WORDS; w += 128) remembered[w] = ctx[w];
    if (threadIdx.x == 0) remembered[KEEP_CTX_WORDS] = 1u + (*wide_ok ? 1u : 0u);
}

__global__ void __launch_bounds__(WB_BLOCK, 4) k_ed25519_verify_check_wide(ProjScratch scr, const void* sig, const u32* __restrict__ ctx,
                                                                          Msgs msgs, size_t n, const u32* __restrict__ wide_base,
                                                                          const u32* __restrict__ wide_key, const u32* __restrict__ wide_ok)
{
    if (!*wide_ok) return;                                 // k_ed25519_verify_check_shared decides this batch
    __shared__ unsigned short cols[2 * WB_COLS * WB_BLOCK];
    const size_t i = (size_t)blockIdx.x * WB_BLOCK + threadIdx.x;
    if (i >= n) return;
    u32 pkw[8], Sw[8], h[8], Rw[8];
#pragma unroll
    for (int j = 0; j < 8; j++) pkw[j] = ctx[j];
    load32(Rw, sig, 2 * i);
    ed_hram(h, Rw, pkw, msgs.ptr(i), msgs.len(i));
    sc_mod(h);
    load32(Sw, sig, 2 * i + 1);                            // raw 256 bits: no s < L check (ed25519_verify.c:308)
    unsigned short* cs = cols + threadIdx.x;
    unsigned short* ch = cols + WB_COLS * WB_BLOCK + threadIdx.x;
    wb_columns(cs, WB_BLOCK, Sw);                          // s + L when even: L * B = O
    const u32 h_even = wb_columns<false>(ch, WB_BLOCK, h);    // h + 1 when even: -A may carry torsion, one -A comes off again
    ge_ext T;                                              // (-A = row 1 of the context, affine: Verify_Init's Z is 1)
    ge_double_base_mult_wide(T, wide_base, cs, wide_key, ch, WB_BLOCK, h_even, ctx + 8 + 32);
    store_proj(scr, n, i, T);
}

// ------------------------------------------------------------------------------------------------
// batched inversion + output encoding
// ------------------------------------------------------------------------------------------------
// Lane j owns elements j, j+m, j+2m, ... (m = number of lanes, so every access stays coalesced) and inverts
// their Z's with ONE exponentiation: prefix products forward, z^(p-2) once, then unwinding backwards
// (Montgomery's trick).  A zero Z (garbage Ed25519 key) must come out as 0 exactly like the reference's z^(p-2)
// does, so zeros are replaced by 1 in the product and their inverse is forced to 0.
// Fin::emit(e, zinv) turns element e's projective value and 1/Z into the operation's output bytes.
struct FinishX25519 {                       // out = canonical(num / den)           (curve25519_dh.c:175-178)
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

// K is a compile-time constant and the loops are unrolled: a lane's K elements live in registers (a lone wave per SIMD has
// the whole register file: 64-thread workgroups, no occupancy to protect), so the loads of all K elements are issued up
// front instead of one dependent round trip per element and per pass.  The K - 1 prefix products a lane needs again on
// the way back stay in registers up to K = 14; at K = 16 they are parked in LDS (15 x 2560 bytes per wave, four waves
// per CU: 150 of the 160 KiB, which nothing else in this kernel uses) -- with all 32 field elements in registers the
// allocator spilled 14-25 of them to scratch.  (`prefix` stays in the signature for the scratch layout's sake.)
template <typename Fin, int K>
__global__ void __launch_bounds__(INV_BLOCK) __attribute__((amdgpu_waves_per_eu(1, 1))) k_batch_invert(const u32* Z, u32* prefix, size_t n, size_t m, Fin fin)
{
    (void)prefix;
    constexpr bool PREFIX_IN_LDS = K > 14;
    __shared__ u32 pre_lds[PREFIX_IN_LDS ? (K - 1) * 10 * INV_BLOCK : 1];
    const size_t j = (size_t)blockIdx.x * INV_BLOCK + threadIdx.x;
    if (j >= m) return;
    fe z[K], pre[PREFIX_IN_LDS ? 1 : K];
    u32 zero_mask = 0;
#pragma unroll
    for (int t = 0; t < K; t++) {
        const size_t e = j + (size_t)t * m;
        if (e < n) soa_load_fe(z[t], Z, n, e);
        else fe_set_u32(z[t], 1);                           // past the end: a factor of one
    }
    fe acc;
#pragma unroll
    for (int t = 0; t < K; t++) {
        zero_mask |= (fe_zero_to_one(z[t]) & 1u) << t;      // z == 0 (mod p) takes no part in the product
        if (t == 0) acc = z[0];
        else fe_mul(acc, acc, z[t]);
        if (t < K - 1) {
            if (PREFIX_IN_LDS) lds_put_fe(pre_lds + t * 10 * INV_BLOCK, INV_BLOCK, threadIdx.x, acc);
            else pre[t] = acc;
        }
    }
    fe inv;
    fe_invert(inv, acc);
#pragma unroll
    for (int t = K - 1; t >= 0; t--) {
        const size_t e = j + (size_t)t * m;
        fe zi;
        if (t > 0) {
            fe p;
            if (PREFIX_IN_LDS) lds_get_fe(p, pre_lds + (t - 1) * 10 * INV_BLOCK, INV_BLOCK, threadIdx.x);
            else p = pre[t - 1];
            fe_mul(zi, inv, p);
            fe_mul(inv, inv, z[t]);
        } else {
            zi = inv;
        }
        const u32 was_zero = ((zero_mask >> t) & 1u) ? 0xffffffffu : 0u;
        fe zero;
        fe_set_u32(zero, 0);
        fe_select(zi, was_zero, zero, zi);
        if (e < n) fin.emit(e, zi);
    }
}

// ------------------------------------------------------------------------------------------------
// unit-test hooks (the counterpart of the reference's ECP_SELF_TEST unit checks,
// test/curve25519_selftest.c:624-741): one lane per input record, operations defined in lanes.cuh
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) k_fe_selftest(void* out, const void* a, const void* b, size_t n, int op)
{
    const size_t i = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    u32 aw[8], bw[8], ow[8];
    load32(aw, a, i);
    load32(bw, b, i);
    fe_selftest_op(ow, aw, bw, op);
    store32(out, i, ow);
}

// op 14 (the division steps on a quad of lanes, safegcd25519.cuh): FOUR lanes per record, all on the same values
__global__ void __launch_bounds__(64) k_fe_selftest_quad(void* out, const void* a, const void* b, size_t n, int op)
{
    const size_t i = (size_t)blockIdx.x * 16 + (threadIdx.x >> 2);
    if (i >= n) return;                                   // (whole quads leave)
    u32 aw[8], bw[8], ow[8];
    load32(aw, a, i);
    load32(bw, b, i);
    fe_selftest_op(ow, aw, bw, op);
    if ((threadIdx.x & 3) == 0) store32(out, i, ow);
}

__global__ void __launch_bounds__(64) k_sc_selftest(void* out, const void* a, const void* b, size_t n, int op)
{
    const size_t i = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    u32 lo[8], hi[8], aw[16], bw[8], ow[8];
    load32(lo, a, 2 * i);
    load32(hi, a, 2 * i + 1);
    load32(bw, b, i);
#pragma unroll
    for (int j = 0; j < 8; j++) { aw[j] = lo[j]; aw[8 + j] = hi[j]; }
    sc_selftest_op(ow, aw, bw, op);
    store32(out, i, ow);
}

__global__ void __launch_bounds__(64) k_fold_selftest(uint8_t* out /* n x 128 */, const void* k, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    u32 kw[8];
    load32(kw, k, i);
    fold_selftest_op(out + 128 * i, kw);
}

// ================================================================================================
// host side
// ================================================================================================
namespace {

using c25519_host::Arr;
using c25519_host::ThreadState;
using c25519_host::aligned16;
using c25519_host::round_up;
using c25519_host::run_batch;
using c25519_host::bad_arg;
using c25519_host::tls;

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
// the wide comb is the default: sign 824 against 643 M/s, key pairs 1110 against 815 M/s at 2^20 (profiles/r05_ab_base_comb.txt)
inline bool base_comb_wide() { return c25519_host::tunable_or(c25519_host::T_BASE_COMB, 1) == 1; }

// *_dev arguments: n in range, pointers 16-byte aligned and -- unless C25519_AMD_NO_PTR_CHECK is set -- device (or
// managed) memory of the CURRENT device: a pointer of another GPU or a host pointer is an error here, not a fault
// inside a kernel.
// the completion word for the LAST kernel of a call of one element, if the caller (host_pipeline.hpp: run_batch on a zero-copy
// call) is going to spin on it; taken at most once per call
DoneWord take_done_word(size_t n)
{
    ThreadState& t = tls();
    if (n != 1 || !t.done_offered || t.done_taken) return DoneWord{ nullptr, 0 };
    t.done_taken = true;
    return DoneWord{ t.done_word, ++t.done_seq };
}

// the two 32-byte records of a one-element call for the kernel's arguments (lanes.cuh: CallWords): only where the "device"
// pointers are this library's own pinned staging, which the host can read (a zero-copy call, host_pipeline.hpp)
CallWords call_words(size_t n, const void* rec0, const void* rec1)
{
    CallWords cw{};
    if (n != 1 || !c25519_host::zero_copy_call()) return cw;
    if (rec0) memcpy(cw.w, rec0, 32);
    if (rec1) memcpy(cw.w + 8, rec1, 32);
    cw.use = 1;
    return cw;
}

int check_dev_args(size_t n, std::initializer_list<const void*> ptrs)
{
    static const bool check_owner_env = getenv("C25519_AMD_NO_PTR_CHECK") == nullptr;
    const bool check_owner = check_owner_env && !c25519_host::zero_copy_call();   // (a tiny *_batch call hands over this library's own pinned staging: host_pipeline.hpp)
    if (n > ((size_t)1 << 31)) return bad_arg("batch too large (n > 2^31)");
    int dev = 0;
    if (check_owner && n) C25519_TRY(hipGetDevice(&dev));
    for (const void* p : ptrs) {
        if (!p) continue;
        if (!aligned16(p)) return bad_arg("device pointers must be 16-byte aligned");
        if (!check_owner || n == 0) continue;
        hipPointerAttribute_t attr;
        if (hipPointerGetAttributes(&attr, p) != hipSuccess) {
            (void)hipGetLastError();
            return bad_arg("*_dev entry points take device pointers (this one is unknown to the HIP runtime)");
        }
        if (attr.type == hipMemoryTypeHost && c25519_host::zero_copy_call()) continue;   // the pinned staging of a tiny *_batch call (host_pipeline.hpp)
        if (attr.type != hipMemoryTypeDevice && attr.type != hipMemoryTypeManaged)
            return bad_arg("*_dev entry points take device pointers (got host memory)");
        if (attr.type == hipMemoryTypeDevice && attr.device != dev)
            return bad_arg("device pointer belongs to another device than the current one");
    }
    return 0;
}


// words of the projective-result part of the scratch for n elements (a, b, z, prefix; 16-byte aligned parts)
inline size_t proj_words(size_t n) { return 4 * round_up(SCR_FE * n, 4); }

ProjScratch carve_proj(u32* base, size_t n)
{
    const size_t part = round_up(SCR_FE * n, 4);
    return ProjScratch{ base, base + part, base + 2 * part, base + 3 * part };
}

// how many elements share one inversion: as many as possible while every SIMD still gets a wave
// (measured at n = 2^20: K = 2 / 4 / 8 / 16 -> 9.52 / 9.39 / 9.33 / 9.29 ms per two-launch X25519 pass)
inline int inversion_k(size_t n)
{
    const long v = c25519_host::tunable(c25519_host::T_INV_K);    // tuning knob, 1..16
    if (v >= 1 && v <= INV_MAX_K) return (int)v;
    size_t k = n / ((size_t)1024 * 64);
    if (k < 1) k = 1;
    if (k > INV_MAX_K) k = INV_MAX_K;
    return (int)k;
}

template <typename Fin>
int launch_invert(const ProjScratch& scr, size_t n, const Fin& fin, hipStream_t stream)
{
    int K = inversion_k(n);
    K = K >= 16 ? 16 : K >= 14 ? 14 : K >= 12 ? 12 : K >= 8 ? 8 : K >= 4 ? 4 : K >= 2 ? 2 : 1;           // the instantiated group sizes
    const size_t m = (n + K - 1) / K;
    const unsigned grid = grid_for(m, INV_BLOCK);
    switch (K) {
        case 16: k_batch_invert<Fin, 16><<<grid, INV_BLOCK, 0, stream>>>(scr.z, scr.prefix, n, m, fin); break;
        case 14: k_batch_invert<Fin, 14><<<grid, INV_BLOCK, 0, stream>>>(scr.z, scr.prefix, n, m, fin); break;
        case 12: k_batch_invert<Fin, 12><<<grid, INV_BLOCK, 0, stream>>>(scr.z, scr.prefix, n, m, fin); break;
        case 8:  k_batch_invert<Fin, 8><<<grid, INV_BLOCK, 0, stream>>>(scr.z, scr.prefix, n, m, fin); break;
        case 4:  k_batch_invert<Fin, 4><<<grid, INV_BLOCK, 0, stream>>>(scr.z, scr.prefix, n, m, fin); break;
        case 2:  k_batch_invert<Fin, 2><<<grid, INV_BLOCK, 0, stream>>>(scr.z, scr.prefix, n, m, fin); break;
        default: k_batch_invert<Fin, 1><<<grid, INV_BLOCK, 0, stream>>>(scr.z, scr.prefix, n, m, fin); break;
    }
    C25519_TRY(hipGetLastError());
    return 0;
}

// scratch of one verification pass: per-lane tables (the larger of the two paths' formats: they never live at the same
// time for one element), projective results of the reference-order path (the fast path keeps its decoded points there),
// the fast path's scalars, flags and slow list
constexpr size_t VERIFY_TABLE_WORDS = FAST_TABLE_WORDS > QTABLE_LIMB_WORDS ? FAST_TABLE_WORDS : QTABLE_LIMB_WORDS;
static_assert(FAST_TABLE_WORDS % 32 == 0 && VERIFY_TABLE_WORDS % 32 == 0, "per-lane tables must keep their rows 128-byte aligned");
inline size_t verify_scalar_words(size_t n) { return round_up(SIGMA_WORDS * n, 4) + 2 * round_up(5 * n, 4) + 5 * round_up(n, 4) + 4; }
inline size_t verify_scratch_bytes(size_t n)
{
    return (n * VERIFY_TABLE_WORDS + proj_words(n) + verify_scalar_words(n)) * sizeof(u32);
}

// fast = true: the lattice path (verify_fast.cuh) decides every element whose key is on the curve and whose short vector
// fits; the reference's order runs for the others in a kernel of its own behind the walk.  fast = false: reference order for everything,
// and Fin decides what leaves it: the verdict, or enc(T) for the test hook.
// what the calling thread's last fast-path verification left behind for c25519_amd_verify_last_slow_elements
struct LastVerify { const u32* count = nullptr; hipStream_t stream = nullptr; int device = -1; unsigned long generation = 0; };
thread_local LastVerify tl_last_verify;

bool verify_coop_for(size_t n);
bool verify_quad_for(size_t n);

// what the calling thread's last ed25519_Verify_Check_* call on this device left behind for c25519_amd_verify_check_last_wide:
// where its "the two wide combs decide this batch" word lives (null: the call never asked)
struct LastCheck { const u32* wide_ok = nullptr; hipStream_t stream = nullptr; int device = -1; unsigned long generation = 0; bool ran = false; };
thread_local LastCheck tl_last_check;

template <typename MakeFin>
int verify_run(const void* sig, const void* pk, Msgs msgs, size_t n, hipStream_t stream, int* verdict, bool fast, MakeFin make_fin)
{
    const u32* tbl = nullptr;
    C25519_RC(base_tables(&tbl, nullptr));
    void* w = nullptr;
    c25519_host::WorkLease lease;
    C25519_RC(lease.acquire(&w, verify_scratch_bytes(n), stream));
    u32* tables = (u32*)w;                                  // first in the slab (hipMalloc: 256-byte aligned): packed rows are
    const ProjScratch scr = carve_proj(tables + n * VERIFY_TABLE_WORDS, n);   // whole 128-byte lines
    const unsigned grid = grid_for(n, ED_BLOCK);
    if (fast) {
        unsigned* report = nullptr;
        C25519_RC(tls().report_word_for(&report, stream));
        FastScratch fs;
        fs.tables = tables;
        fs.sigma = tables + n * VERIFY_TABLE_WORDS + proj_words(n);
        fs.rho = fs.sigma + round_up(SIGMA_WORDS * n, 4);
        fs.tau = fs.rho + round_up(5 * n, 4);
        fs.flags = fs.tau + round_up(5 * n, 4);
        fs.slow_list = fs.flags + round_up(n, 4);
        fs.order = fs.slow_list + round_up(n, 4);
        fs.slow_count = fs.order + round_up(n, 4);
        fs.pflags = fs.slow_count + 4;
        fs.slow_report = report;
        {   // test knob: a lower cap sends ordinary signatures down the over-long-vector branch (slow list, reference order)
            const long cap = c25519_host::tunable_or(c25519_host::T_VERIFY_LAT_CAP_BITS, LAT_CAP_BITS);
            fs.lat_cap_bits = cap >= 100 && cap < LAT_CAP_BITS ? (int)cap : LAT_CAP_BITS;
        }
        if (!verify_quad_for(n) && verify_coop_for(n)) {   // a few elements: one launch, three waves per element
            C25519_TRY(hipMemsetAsync(fs.slow_count, 0, 3 * sizeof(u32), stream));
            k_ed25519_verify_one_per_group<<<(unsigned)n, 192, 0, stream>>>(fs, verdict, sig, pk, msgs, n, tbl);
            C25519_TRY(hipGetLastError());
        } else if (verify_quad_for(n)) {                   // four lanes per element walk; scalars and points side by side in one launch
            const unsigned sb = grid_for(n, FS_BLOCK);
            k_ed25519_verify_quad_prep<<<sb + grid_for(2 * n, ED_BLOCK), ED_BLOCK, 0, stream>>>(fs, sig, pk, msgs, n, sb);
            C25519_TRY(hipGetLastError());
            k_ed25519_verify_quad_walk<<<grid_for(n, QW_BLOCK / 4), QW_BLOCK, 0, stream>>>(fs, verdict, n, tbl);
            C25519_TRY(hipGetLastError());
        } else {
            k_ed25519_verify_fast_scalars<<<grid_for(n, FS_BLOCK), FS_BLOCK, 0, stream>>>(fs, sig, pk, msgs, n);
            C25519_TRY(hipGetLastError());
            k_ed25519_verify_fast_points<<<grid_for(2 * n, ED_BLOCK), ED_BLOCK, 0, stream>>>(fs, sig, pk, n);
            C25519_TRY(hipGetLastError());
            k_ed25519_verify_fast_walk<<<grid_for(n, WALK_BLOCK), WALK_BLOCK, 0, stream>>>(fs, verdict, n, tbl);
            C25519_TRY(hipGetLastError());
        }
        k_ed25519_verify_slow<<<grid, ED_BLOCK, 0, stream>>>(fs, verdict, sig, pk, msgs, tbl, take_done_word(n));
        C25519_TRY(hipGetLastError());
        tl_last_verify.count = report; tl_last_verify.stream = stream;
        tl_last_verify.generation = tls().generation;       // the report word and the stream die with the thread's slabs
        (void)hipGetDevice(&tl_last_verify.device);
        return lease.release();
    }
    tl_last_verify = LastVerify();
    k_ed25519_verify_init<QTableLimbs><<<grid, ED_BLOCK, 0, stream>>>(pk, n, tables, VERIFY_TABLE_WORDS);
    C25519_TRY(hipGetLastError());
    k_ed25519_verify_check<QTableLimbs><<<grid, ED_BLOCK, 0, stream>>>(scr, sig, pk, msgs, n, tbl, tables, VERIFY_TABLE_WORDS);
    C25519_TRY(hipGetLastError());
    C25519_RC(launch_invert(scr, n, make_fin(scr), stream));
    return lease.release();
}

// lanes per X25519 workgroup for a batch of n: the widest shape that still puts a wave on every SIMD the batch can reach
// (256 CUs x 4 SIMDs; 2^16 elements are 1024 waves).  profiles/r03_batch_sweep.txt has both shapes side by side.
int x25519_block_for(size_t n)
{
    n = std::max(n, c25519_host::batch_shape_hint());         // a piece of a pipelined *_batch call: the whole call counts
    if (n <= ((size_t)1 << 16)) return 64;
    if (n <= ((size_t)1 << 17)) return 128;
    if (n <= ((size_t)1 << 18)) return 256;
    return XF_BLOCK;
}

// a call of a few elements -- the reference's single-call prototypes are a batch of one -- runs ONE operation per wave
// (k_x25519_coop): ~5 x less latency than one operation per lane, at ~12 x the instructions per operation, so only while
// the waves still find idle SIMDs.  Tunable COOP_MAX = the largest such batch (A/B and test knob; 0 = never; at most 2^20:
// one workgroup per element).
bool coop_for(size_t n, size_t dflt)
{
    const long v = c25519_host::tunable(c25519_host::T_COOP_MAX);
    const size_t max = v == c25519_host::T_UNSET ? dflt : (size_t)std::min<long>(std::max<long>(v, 0), 1L << 20);
    return n <= max && c25519_host::batch_shape_hint() <= max;
}
// crossovers measured on MI355X (tools/small_batch_sweep.py, profiles/r04_small_batch_sweep.txt): the ladder one per wave
// wins up to 4096 elements (0.49 against 0.66 ms), the fixed-base operations up to 2048 (0.10-0.16 against 0.15-0.19 ms),
// verification (three waves per element, profiles/r05_small_batch_sweep.txt) up to 2048
bool x25519_coop_for(size_t n) { return coop_for(n, 4096); }
// ... two waves per element while every wave still finds a SIMD of its own: 167 against 179 us for one element, 193 against 201 for
// 512, 213 against 216 for 1024 (profiles/r05_small_batch_sweep.txt; tunable LADDER2_MAX; a per-wave call in any case)
bool x25519_two_waves_for(size_t n)
{
    const long v = c25519_host::tunable(c25519_host::T_LADDER2_MAX);
    const size_t max = v == c25519_host::T_UNSET ? 512 : (size_t)std::min<long>(std::max<long>(v, 0), 1L << 20);
    return n <= max && x25519_coop_for(n);
}
// four lanes per element (k_x25519_quad): between the per-wave kernels and the batches that give every SIMD a wave of one-lane
// elements.  X25519: the quad's step is 679 instructions against the lane's 1246, so up to 2^14 elements (1024 quad-waves, one per
// SIMD) a call takes 0.34 ms instead of 0.71 (23 / 48 M/s at 2^13 / 2^14 against 11.6 / 23.1); two quad-waves per SIMD (2^15
// elements) still beat the 512 one-lane waves, 0.62 against 0.71 ms; below ~3600 elements a wave per element is faster.
// Tunables QUAD_MIN / QUAD_MAX (tools/mid_batch_sweep.py, profiles/r06_mid_batch_sweep.txt).
bool quad_for(size_t n, size_t dflt_min, size_t dflt_max)
{
    const long lo = c25519_host::tunable(c25519_host::T_QUAD_MIN), hi = c25519_host::tunable(c25519_host::T_QUAD_MAX);
    const size_t mn = lo == c25519_host::T_UNSET ? dflt_min : (size_t)std::max<long>(lo, 0);
    const size_t mx = hi == c25519_host::T_UNSET ? dflt_max : (size_t)std::min<long>(std::max<long>(hi, 0), 1L << 24);
    const size_t m = std::max(n, c25519_host::batch_shape_hint());     // a piece of a pipelined *_batch call: the whole call counts
    return m > mn && m <= mx;
}
bool x25519_quad_for(size_t n) { return quad_for(n, 3584, (size_t)1 << 15); }
bool verify_quad_for(size_t n) { return quad_for(n, 1024, (size_t)1 << 15); }      // k_ed25519_verify_quad_prep + _quad_walk: 0.30-0.31 ms up to 2^14, 0.48 at 2^15 (one-lane kernels: 0.52-0.63)
// the fixed-base operations on quads (k_ed25519_*_quad; over the wide comb, without a blinding context): one chain of 53-89 us up to
// 2^14 elements (one quad-wave per SIMD) against 81 us for 1024 per-wave signatures and the one-lane path's three launches
// (134-144 us at 2^15 / 2^16); profiles/r06_mid_batch_sweep.txt
bool fixed_base_quad_for(size_t n) { return quad_for(n, 1024, (size_t)1 << 14); }
bool fixed_base_coop_for(size_t n) { return coop_for(n, 2048); }
bool verify_coop_for(size_t n) { return coop_for(n, 2048); }       // three waves per element: 0.13-0.55 against 0.60 ms (1.02 at 4096)

// a batch that fills the chip runs the ladder and the shared inversion as two launches (k_x25519_ladder's comment);
// tunable XF_SPLIT = 0 / 1 forces either shape (A/B knob)
bool x25519_split_for(size_t n)
{
    const long v = c25519_host::tunable(c25519_host::T_XF_SPLIT);
    if (v != c25519_host::T_UNSET) return v != 0;
    return std::max(n, c25519_host::batch_shape_hint()) > ((size_t)1 << 16);   // measured at the sustained clock: two launches win from 2^17 up (3 / 2 / 1.2 % at 2^17 / 2^18 / 2^20), one launch by 1 % below
}

template <int BLOCK>
void x25519_launch(void* out, const void* pk, void* sk, size_t n, hipStream_t stream)
{
    if (pk) k_x25519_fused<false, BLOCK><<<grid_for(n, BLOCK), BLOCK, 0, stream>>>(out, pk, sk, n);
    else    k_x25519_fused<true, BLOCK><<<grid_for(n, BLOCK), BLOCK, 0, stream>>>(out, pk, sk, n);
}

}  // namespace

extern "C" {

const char* c25519_amd_version(void) { return "curve25519_amd 0.7 (gfx950)"; }
const char* c25519_amd_last_error(void) { return c25519_host::last_error().c_str(); }

int c25519_amd_device_count(void)
{
    C25519_API_CALL_OR(0);
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}

int c25519_amd_host_register(void* p, size_t bytes)
{
    C25519_API_CALL();
    if (!p || !bytes) return bad_arg("null pointer or empty range");
    // page locking works on whole pages: a buffer that shares a page with another allocation would get that neighbour
    // locked, and unlocked, with it (the runtime aborts on the second unregister) -- so only whole pages are accepted
    if ((reinterpret_cast<uintptr_t>(p) & 4095u) || (bytes & 4095u)) return bad_arg("host_register: the buffer must start on a 4 KiB page and cover whole pages");
    C25519_TRY(hipHostRegister(p, bytes, hipHostRegisterDefault));
    return 0;
}

int c25519_amd_host_unregister(void* p)
{
    C25519_API_CALL();
    if (!p) return bad_arg("null pointer");
    C25519_TRY(hipHostUnregister(p));
    return 0;
}

// tuning / A-B knobs (capi_common.hpp: Tunable).  name = the part behind C25519_AMD_ of the environment variable that
// initialises the knob; value < 0 restores the library's built-in choice.
int c25519_amd_tunable_set(const char* name, long value)
{
    if (!name) return bad_arg("null pointer");
    for (int i = 0; i < c25519_host::T_COUNT; i++)
        if (!strcmp(name, c25519_host::tunable_names()[i])) {
            c25519_host::tunable_table()[i].store(value < 0 ? c25519_host::T_UNSET : value, std::memory_order_relaxed);
            return 0;
        }
    return bad_arg("c25519_amd_tunable_set: no such knob");
}

long c25519_amd_tunable_get(const char* name)
{
    if (name)
        for (int i = 0; i < c25519_host::T_COUNT; i++)
            if (!strcmp(name, c25519_host::tunable_names()[i])) return c25519_host::tunable((c25519_host::Tunable)i);
    return -2;
}

int c25519_amd_usable_cpus(void) { return c25519_host::usable_cpus(); }

int c25519_amd_set_device(int device)
{
    C25519_API_CALL();
    C25519_TRY(hipSetDevice(device));
    return 0;
}

#ifdef C25519_CYCLE_PROBE
// measurement builds only: where the waves of k_x25519_fused write their stamps (PROBE_WORDS u64 per wave), or null
int c25519_amd_probe_set(void* buf)
{
    C25519_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_cycle_probe), &buf, sizeof buf));
    return 0;
}
int c25519_amd_probe_words(void) { return PROBE_WORDS; }
#endif

// frees the calling thread's streams, staging buffers (zeroed first) and work scratch
void c25519_amd_thread_release(void)
{
    if (!c25519_host::runtime_alive().load()) return;                   // exit() has begun: the process' memory goes with it
    C25519_API_CALL_OR((void)0);
    c25519_host::helper_pool_slot().reset();              // the pipeline's parked helper threads
    tls().release();
}

// ---- device-pointer entry points ----------------------------------------------------------------

static int x25519_dev(void* out, const void* pk, void* sk, size_t n, hipStream_t stream)
{
    if (x25519_quad_for(n)) {                                 // four lanes per element
        const unsigned grid = grid_for(n, quad::ELEMS_PER_WAVE);
        if (pk) k_x25519_quad<false><<<grid, 64, 0, stream>>>(out, pk, sk, n);
        else    k_x25519_quad<true><<<grid, 64, 0, stream>>>(out, pk, sk, n);
        C25519_TRY(hipGetLastError());
        return 0;
    }
    if (x25519_coop_for(n)) {
        const CallWords cw = call_words(n, pk, sk);
        if (pk && x25519_two_waves_for(n)) k_x25519_coop2<<<(unsigned)n, 128, 0, stream>>>(out, pk, sk, n, take_done_word(n), cw);
        else if (pk) k_x25519_coop<false><<<(unsigned)n, 64, 0, stream>>>(out, pk, sk, n, take_done_word(n), cw);
        else    k_x25519_coop<true><<<(unsigned)n, 64, 0, stream>>>(out, pk, sk, n, take_done_word(n), cw);
        C25519_TRY(hipGetLastError());
        return 0;
    }
    if (x25519_split_for(n)) {
        void* w = nullptr;
        c25519_host::WorkLease lease;
        C25519_RC(lease.acquire(&w, proj_words(n) * sizeof(u32), stream));
        const ProjScratch scr = carve_proj((u32*)w, n);
        if (pk) k_x25519_ladder<false><<<grid_for(n, XL_BLOCK), XL_BLOCK, 0, stream>>>(scr.a, scr.z, pk, sk, n);
        else    k_x25519_ladder<true><<<grid_for(n, XL_BLOCK), XL_BLOCK, 0, stream>>>(scr.a, scr.z, pk, sk, n);
        C25519_TRY(hipGetLastError());
        C25519_RC(launch_invert(scr, n, FinishX25519{ scr.a, out, n }, stream));
        return lease.release();
    }
    switch (x25519_block_for(n)) {
    case 64:  x25519_launch<64>(out, pk, sk, n, stream); break;
    case 128: x25519_launch<128>(out, pk, sk, n, stream); break;
    case 256: x25519_launch<256>(out, pk, sk, n, stream); break;
    default:  x25519_launch<XF_BLOCK>(out, pk, sk, n, stream); break;
    }
    C25519_TRY(hipGetLastError());
    return 0;
}

int curve25519_dh_CreateSharedKey_dev(void* shared, const void* pk, void* sk, size_t n, void* stream)
{
    C25519_API_CALL();
    if (!shared || !pk || !sk) return bad_arg("null pointer");
    if (int rc = check_dev_args(n, { shared, pk, sk })) return rc;
    if (n == 0) return 0;
    return x25519_dev(shared, pk, sk, n, (hipStream_t)stream);
}

int curve25519_dh_CalculatePublicKey_dev(void* pk, void* sk, size_t n, void* stream)
{
    C25519_API_CALL();
    if (!pk || !sk) return bad_arg("null pointer");
    if (int rc = check_dev_args(n, { pk, sk })) return rc;
    if (n == 0) return 0;
    return x25519_dev(pk, nullptr, sk, n, (hipStream_t)stream);
}

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

static int keypair_dev(void* pub, void* priv, const void* sk, const void* blinding, size_t n, hipStream_t stream)
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
            k_ed25519_keypair_coop<true><<<(unsigned)n, 64, 0, stream>>>(pub, priv, sk, n, wide, (const u32*)blinding, take_done_word(n));
        } else k_ed25519_keypair_coop<false><<<(unsigned)n, 64, 0, stream>>>(pub, priv, sk, n, tbl, nullptr, take_done_word(n));
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

static int sign_dev(void* sig, const void* priv, const void* blinding, Msgs msgs, size_t n, hipStream_t stream)
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
        if (wide_comb) {
            const u32* wide = nullptr;
            C25519_RC(wide_tables(&wide));
            k_ed25519_sign_coop<true><<<(unsigned)n, 64, 0, stream>>>(sig, priv, msgs, n, wide, (const u32*)blinding, take_done_word(n));
        } else k_ed25519_sign_coop<false><<<(unsigned)n, 64, 0, stream>>>(sig, priv, msgs, n, tbl, nullptr, take_done_word(n));
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

size_t ed25519_VerifySignature_scratch_bytes(size_t n) { return verify_scratch_bytes(n); }

static int verify_dev(void* verdict, const void* sig, const void* pk, Msgs msgs, size_t n, hipStream_t stream)
{
    // tunable VERIFY_REFERENCE_ORDER = 1: every element through the reference-order kernels -- Verify_Init's 4-fold table per
    // key, then the 4-fold + 8-fold walk of ed25519_verify.c:243-280: BASELINE.json configs[3] as worded (A/B and test knob)
    const bool fast = c25519_host::tunable_or(c25519_host::T_VERIFY_REFERENCE_ORDER, 0) == 0;
    if (int rc = check_dev_args(n, { verdict, sig, pk })) return rc;
    if (n == 0) return 0;
    return verify_run(sig, pk, msgs, n, stream, (int*)verdict, fast,
                      [&](const ProjScratch& scr) { return FinishVerify{ scr.a, scr.b, sig, (int*)verdict, n }; });
}

// test hook: enc(T) instead of the verdict (what Verify_Check compares with enc(R)); device pointers
int c25519_amd_verify_point_dev(void* out, const void* sig, const void* pk, const void* msg, size_t msg_size, size_t n,
                                void* stream)
{
    C25519_API_CALL();
    if (!out || !sig || !pk || (!msg && msg_size)) return bad_arg("null pointer");
    if (int rc = check_dev_args(n, { out, sig, pk })) return rc;
    if (n == 0) return 0;
    return verify_run(sig, pk, Msgs{ (const uint8_t*)msg, msg_size, nullptr }, n, (hipStream_t)stream, nullptr, false,
                      [&](const ProjScratch& scr) { return FinishPack{ scr.a, scr.b, out, n, 1, 0, nullptr, 0, 0 }; });
}

// how many elements of the calling thread's last ed25519_VerifySignature_* call on this device went through the
// reference-order kernel instead of the lattice path (-1: no fast-path verification to report).  Synchronises.
long c25519_amd_verify_last_slow_elements(void)
{
    C25519_API_CALL_OR(-1);
    const LastVerify& lv = tl_last_verify;
    int dev = -1;
    if (!lv.count || hipGetDevice(&dev) != hipSuccess || dev != lv.device) return -1;
    if (lv.generation != tls().generation) return -1;       // c25519_amd_thread_release() / a device switch freed what lv points at
    if (hipStreamSynchronize(lv.stream) != hipSuccess) return -1;
    u32 c = 0;
    if (hipMemcpy(&c, lv.count, sizeof c, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return (long)c;
}

int ed25519_VerifySignature_dev(void* verdict, const void* sig, const void* pk, const void* msg, size_t msg_size,
                                size_t n, void* stream)
{
    C25519_API_CALL();
    if (!verdict || !sig || !pk || (!msg && msg_size)) return bad_arg("null pointer");
    return verify_dev(verdict, sig, pk, Msgs{ (const uint8_t*)msg, msg_size, nullptr }, n, (hipStream_t)stream);
}

int ed25519_VerifySignature_ragged_dev(void* verdict, const void* sig, const void* pk, const void* msgs,
                                       const uint64_t* offsets, size_t n, void* stream)
{
    C25519_API_CALL();
    if (!verdict || !sig || !pk || !offsets) return bad_arg("null pointer");
    return verify_dev(verdict, sig, pk, Msgs{ (const uint8_t*)msgs, 0, (const unsigned long long*)offsets }, n,
                      (hipStream_t)stream);
}

// two-phase verification on the device: contexts are 2080-byte records (pk || 16 x 128-byte canonical rows),
// the reference's EDP_SIGV_CTX size and row order.
int ed25519_Verify_Init_dev(void* ctx, const void* pk, size_t n, void* stream)
{
    C25519_API_CALL();
    if (!ctx || !pk) return bad_arg("null pointer");
    if (int rc = check_dev_args(n, { ctx, pk })) return rc;
    if (n == 0) return 0;
    if (coop_for(n, 1024))                                  // a few keys: one per wave (which also copies its key into the context)
        k_ed25519_verify_init_coop<<<(unsigned)n, 64, 0, (hipStream_t)stream>>>(pk, n, (u32*)ctx + 8, 2080 / 4, take_done_word(n));
    else {
        C25519_TRY(hipMemcpy2DAsync(ctx, 2080, pk, 32, 32, n, hipMemcpyDeviceToDevice, (hipStream_t)stream));
        k_ed25519_verify_init<QTableCanon><<<grid_for(n, ED_BLOCK), ED_BLOCK, 0, (hipStream_t)stream>>>(
            pk, n, (u32*)ctx + 8, 2080 / 4);
    }
    C25519_TRY(hipGetLastError());
    return 0;
}

int ed25519_Verify_Check_dev(void* verdict, const void* ctx, const void* sig, const void* msg, size_t msg_size,
                             size_t n, void* stream_)
{
    C25519_API_CALL();
    if (!verdict || !ctx || !sig || (!msg && msg_size)) return bad_arg("null pointer");
    if (int rc = check_dev_args(n, { verdict, ctx, sig })) return rc;
    if (n == 0) return 0;
    hipStream_t stream = (hipStream_t)stream_;
    const u32* tbl = nullptr;
    C25519_RC(base_tables(&tbl, nullptr));
    // a big batch under one key: both scalars over wide combs, if the context is Verify_Init's own and the key is on the
    // curve (k_ed25519_verify_check_wide); decided on the device, the reference-order kernel behind it takes the batch otherwise.
    // Building the key's comb (0.6 ms) pays from ONE_KEY_WIDE signatures per call (2^16); a comb that is REMEMBERED -- one
    // Verify_Init, many Verify_Check calls, ed25519_verify.c:282-286 -- costs nothing, so every call above the per-wave kernels'
    // range asks the device whether its context is the remembered one (one block, 2080 bytes out of L2) and walks the combs if so.
    const long wide_from = c25519_host::tunable_or(c25519_host::T_ONE_KEY_WIDE, 1 << 16);      // (read once per call)
    const bool small = coop_for(n, 1024);
    const bool build = wide_from != 0 && n >= (size_t)wide_from;
    const bool reuse = !build && wide_from != 0 && !small && tls().has_keep();
    const bool try_wide = build || reuse;
    tl_last_check = LastCheck();
    tl_last_check.ran = true;
    if (!try_wide && small) {                               // a few pairs: one per wave, the reference's order
        k_ed25519_verify_check_coop<<<(unsigned)n, 64, 0, stream>>>((int*)verdict, sig, (const u32*)ctx,
                                                                    Msgs{ (const uint8_t*)msg, msg_size, nullptr }, n, tbl, take_done_word(n));
        C25519_TRY(hipGetLastError());
        return 0;
    }
    void* w = nullptr;
    c25519_host::WorkLease lease;
    C25519_RC(lease.acquire(&w, (proj_words(n) + 4) * sizeof(u32), stream));
    const ProjScratch scr = carve_proj((u32*)w, n);
    u32* wide_ok = nullptr;
    c25519_host::KeepLease keep_lease;                      // records the kept buffer's event however this call leaves
    if (try_wide) {
        const u32* wide_base = nullptr;
        C25519_RC(wide_tables(&wide_base));
        // the key's comb and the context it was built for live in a buffer of the calling thread that outlives the call
        // (ThreadState::keep): the next call with the same context bytes finds them there.  The verdict on THIS call's context
        // (wide_ok) is the call's own: a word of its work scratch.
        void* keep = nullptr;
        bool fresh = false;
        constexpr size_t KEEP_WORDS = WB_TBL_WORDS + 16 * 32 + KEEP_CTX_WORDS + 1 + 3;
        C25519_RC(keep_lease.acquire(&keep, KEEP_WORDS * sizeof(u32), stream, &fresh));
        u32* wide_key = (u32*)keep;
        u32* check_rows = wide_key + WB_TBL_WORDS;
        u32* remembered = check_rows + 16 * 32;
        wide_ok = (u32*)w + proj_words(n);                  // (16-byte aligned: proj_words is a multiple of 4)
        tl_last_check.wide_ok = wide_ok; tl_last_check.stream = stream; tl_last_check.generation = tls().generation;
        (void)hipGetDevice(&tl_last_check.device);
        k_ed25519_verify_ctx_prepare<<<build ? 1 + WB_NT * WB_ROWS / 128 : 1, 128, 0, stream>>>(wide_key, check_rows, wide_ok, (const u32*)ctx, remembered, build ? 1 : 0);
        C25519_TRY(hipGetLastError());
        if (build) {
            k_ed25519_verify_ctx_remember<<<1, 128, 0, stream>>>(remembered, (const u32*)ctx, wide_ok);
            C25519_TRY(hipGetLastError());
        }
        k_ed25519_verify_check_wide<<<grid_for(n, WB_BLOCK), WB_BLOCK, 0, stream>>>(
            scr, sig, (const u32*)ctx, Msgs{ (const uint8_t*)msg, msg_size, nullptr }, n, wide_base, wide_key, wide_ok);
        C25519_TRY(hipGetLastError());
    }
    k_ed25519_verify_check_shared<<<grid_for(n, ED_BLOCK), ED_BLOCK, 0, stream>>>(
        scr, sig, (const u32*)ctx, Msgs{ (const uint8_t*)msg, msg_size, nullptr }, n, tbl, wide_ok);
    C25519_TRY(hipGetLastError());
    C25519_RC(launch_invert(scr, n, FinishVerify{ scr.a, scr.b, sig, (int*)verdict, n }, stream));
    C25519_RC(keep_lease.release());
    return lease.release();
}

// test / accounting hook: did the calling thread's last ed25519_Verify_Check_* call on this device walk the two wide combs (1), or
// did the reference-order kernel decide it (0: the call did not ask -- too small, no remembered comb, ONE_KEY_WIDE = 0 -- or the
// device said no: another context than the remembered one, a context that is not Verify_Init's, an off-curve key)?  -1: no such
// call to report.  Synchronises with that call's stream.  (A *_batch call of several pieces reports its last piece.)
long c25519_amd_verify_check_last_wide(void)
{
    C25519_API_CALL_OR(-1);
    const LastCheck& lc = tl_last_check;
    if (!lc.ran) return -1;
    if (!lc.wide_ok) return 0;
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev != lc.device || lc.generation != tls().generation) return -1;
    if (hipStreamSynchronize(lc.stream) != hipSuccess) return -1;
    u32 v = 0;
    if (hipMemcpy(&v, lc.wide_ok, sizeof v, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return v ? 1 : 0;
}

// ---- unit-test hooks (host pointers) ---------------------------------------------------------------
int c25519_amd_fe_selftest(unsigned char* out, const unsigned char* a, const unsigned char* b, size_t n, int op)
{
    if (!out || !a || !b) return bad_arg("null pointer");
    if (n == 0) return 0;
    return run_batch(n, { Arr{ a, nullptr, 32 }, Arr{ b, nullptr, 32 }, Arr{ nullptr, out, 32 } },
                     [&](void** d, size_t c, size_t, hipStream_t st) -> int {
                         if (op == 14) k_fe_selftest_quad<<<grid_for(c, 16), 64, 0, st>>>(d[2], d[0], d[1], c, op);
                         else k_fe_selftest<<<grid_for(c, 64), 64, 0, st>>>(d[2], d[0], d[1], c, op);
                         C25519_TRY(hipGetLastError());
                         return 0;
                     });
}

int c25519_amd_sc_selftest(unsigned char* out, const unsigned char* a, const unsigned char* b, size_t n, int op)
{
    if (!out || !a || !b) return bad_arg("null pointer");
    if (n == 0) return 0;
    return run_batch(n, { Arr{ a, nullptr, 64 }, Arr{ b, nullptr, 32 }, Arr{ nullptr, out, 32 } },
                     [&](void** d, size_t c, size_t, hipStream_t st) -> int {
                         k_sc_selftest<<<grid_for(c, 64), 64, 0, st>>>(d[2], d[0], d[1], c, op);
                         C25519_TRY(hipGetLastError());
                         return 0;
                     });
}

int c25519_amd_fold_selftest(unsigned char* out, const unsigned char* k, size_t n)
{
    if (!out || !k) return bad_arg("null pointer");
    if (n == 0) return 0;
    return run_batch(n, { Arr{ k, nullptr, 32 }, Arr{ nullptr, out, 128 } },
                     [&](void** d, size_t c, size_t, hipStream_t st) -> int {
                         k_fold_selftest<<<grid_for(c, 64), 64, 0, st>>>((uint8_t*)d[1], d[0], c);
                         C25519_TRY(hipGetLastError());
                         return 0;
                     });
}

int c25519_amd_base_table(unsigned char* out)
{
    C25519_API_CALL();
    if (!out) return bad_arg("null pointer");
    const u32* bytes = nullptr;
    if (int rc = base_tables(nullptr, &bytes)) return rc;
    C25519_TRY(hipMemcpy(out, bytes, 256 * 96, hipMemcpyDeviceToHost));
    return 0;
}

// ---- host-pointer entry points: stage, run the *_dev form, copy back (run_batch above) ---------------

int curve25519_dh_CreateSharedKey_batch(unsigned char* shared, const unsigned char* pk, unsigned char* sk, size_t n)
{
    if (!shared || !pk || !sk) return bad_arg("null pointer");
    if (n == 0) return 0;
    return run_batch(n, { Arr{ pk, nullptr, 32 }, Arr{ sk, sk, 32 }, Arr{ nullptr, shared, 32 } },
                     [&](void** d, size_t c, size_t, hipStream_t st) -> int {
                         return curve25519_dh_CreateSharedKey_dev(d[2], d[0], d[1], c, st);
                     });
}

static int public_batch(unsigned char* pk, unsigned char* sk, size_t n, bool fast)
{
    if (!pk || !sk) return bad_arg("null pointer");
    if (n == 0) return 0;
    return run_batch(n, { Arr{ sk, sk, 32 }, Arr{ nullptr, pk, 32 } },
                     [&](void** d, size_t c, size_t, hipStream_t st) -> int {
                         return fast ? curve25519_dh_CalculatePublicKey_fast_dev(d[1], d[0], c, st)
                                     : curve25519_dh_CalculatePublicKey_dev(d[1], d[0], c, st);
                     });
}

int curve25519_dh_CalculatePublicKey_batch(unsigned char* pk, unsigned char* sk, size_t n) { return public_batch(pk, sk, n, false); }
int curve25519_dh_CalculatePublicKey_fast_batch(unsigned char* pk, unsigned char* sk, size_t n) { return public_batch(pk, sk, n, true); }

// the blinding context of a host-pointer call: 192 bytes uploaded once per call into the thread's scratch lane
static int upload_blinding(void** dctx, const void* blinding)
{
    ThreadState& t = tls();
    C25519_RC(t.ensure());
    // a caller signs many times with one context (the reference's C++ wrapper keeps two static ones, C++/ed25519.cpp): it
    // is uploaded when its bytes differ from what this thread uploaded last, not with a synchronous copy per call
    if (!t.bctx) C25519_TRY(hipMalloc(&t.bctx, 4 * BLIND_WORDS));
    if (!t.bctx_valid || memcmp(t.bctx_host, blinding, 4 * BLIND_WORDS) != 0) {
        t.bctx_valid = false;
        C25519_RC(c25519_host::upload_now(t.bctx, blinding, 4 * BLIND_WORDS));
        memcpy(t.bctx_host, blinding, 4 * BLIND_WORDS);
        t.bctx_valid = true;
    }
    *dctx = t.bctx;
    return 0;
}

static int keypair_batch(unsigned char* pub, unsigned char* priv, const void* blinding, const unsigned char* sk, size_t n)
{
    C25519_API_CALL();
    if (!pub || !priv || !sk) return bad_arg("null pointer");
    if (n == 0) return 0;
    void* dctx = nullptr;
    if (blinding) C25519_RC(upload_blinding(&dctx, blinding));
    return run_batch(n, { Arr{ sk, nullptr, 32 }, Arr{ nullptr, pub, 32 }, Arr{ nullptr, priv, 64 } },
                     [&](void** d, size_t c, size_t, hipStream_t st) -> int {
                         return keypair_dev(d[1], d[2], d[0], dctx, c, st);
                     });
}

int ed25519_CreateKeyPair_batch(unsigned char* pub, unsigned char* priv, const unsigned char* sk, size_t n)
{
    return keypair_batch(pub, priv, nullptr, sk, n);
}

int ed25519_CreateKeyPair_blinded_batch(unsigned char* pub, unsigned char* priv, const void* blinding,
                                        const unsigned char* sk, size_t n)
{
    if (!blinding) return bad_arg("null blinding context");
    return keypair_batch(pub, priv, blinding, sk, n);
}

static int sign_batch(unsigned char* sig, const unsigned char* priv, const void* blinding, const unsigned char* msg,
                      size_t msg_size, size_t n)
{
    C25519_API_CALL();
    if (!sig || !priv || (!msg && msg_size)) return bad_arg("null pointer");
    if (n == 0) return 0;
    void* dctx = nullptr;
    if (blinding) C25519_RC(upload_blinding(&dctx, blinding));
    return run_batch(n, { Arr{ priv, nullptr, 64 }, Arr{ msg, nullptr, msg_size }, Arr{ nullptr, sig, 64 } },
                     [&](void** d, size_t c, size_t, hipStream_t st) -> int {
                         return sign_dev(d[2], d[0], dctx, Msgs{ (const uint8_t*)d[1], msg_size, nullptr }, c, st);
                     });
}

int ed25519_SignMessage_batch(unsigned char* sig, const unsigned char* priv, const unsigned char* msg,
                              size_t msg_size, size_t n)
{
    return sign_batch(sig, priv, nullptr, msg, msg_size, n);
}

int ed25519_SignMessage_blinded_batch(unsigned char* sig, const unsigned char* priv, const void* blinding,
                                      const unsigned char* msg, size_t msg_size, size_t n)
{
    if (!blinding) return bad_arg("null blinding context");
    return sign_batch(sig, priv, blinding, msg, msg_size, n);
}

int ed25519_VerifySignature_batch(int* verdict, const unsigned char* sig, const unsigned char* pk,
                                  const unsigned char* msg, size_t msg_size, size_t n)
{
    if (!verdict || !sig || !pk || (!msg && msg_size)) return bad_arg("null pointer");
    if (n == 0) return 0;
    return run_batch(n, { Arr{ sig, nullptr, 64 }, Arr{ pk, nullptr, 32 }, Arr{ msg, nullptr, msg_size },
                          Arr{ nullptr, verdict, sizeof(int) } },
                     [&](void** d, size_t c, size_t, hipStream_t st) -> int {
                         return ed25519_VerifySignature_dev(d[3], d[0], d[1], d[2], msg_size, c, st);
                     });
}

// ragged messages: message i is msgs[offsets[i] .. offsets[i+1]); offsets has n+1 entries (host memory).
// One piece: the message bytes and the offsets are uploaded whole.
static int ragged_upload(ThreadState& t, void** d_msgs, void** d_off, const unsigned char* msgs, const uint64_t* offsets,
                         size_t n)
{
    const int L = ThreadState::LANES - 1;
    C25519_RC(t.reserve_dev(L, 3, (size_t)offsets[n]));
    C25519_RC(t.reserve_dev(L, 4, sizeof(uint64_t) * (n + 1)));
    *d_msgs = t.dbuf[L][3];
    *d_off = t.dbuf[L][4];
    if (offsets[n]) C25519_TRY(hipMemcpyAsync(*d_msgs, msgs, (size_t)offsets[n], hipMemcpyHostToDevice, t.stream[L]));
    C25519_TRY(hipMemcpyAsync(*d_off, offsets, sizeof(uint64_t) * (n + 1), hipMemcpyHostToDevice, t.stream[L]));
    return 0;
}

int ed25519_SignMessage_ragged_batch(unsigned char* sig, const unsigned char* priv, const unsigned char* msgs,
                                     const uint64_t* offsets, size_t n)
{
    C25519_API_CALL();
    if (!sig || !priv || !offsets) return bad_arg("null pointer");
    if (n == 0) return 0;
    ThreadState& t = tls();
    C25519_RC(t.ensure());
    const int L = ThreadState::LANES - 1;
    hipStream_t st = t.stream[L];
    void *d_msgs, *d_off;
    C25519_RC(ragged_upload(t, &d_msgs, &d_off, msgs, offsets, n));
    C25519_RC(t.reserve_dev(L, 0, 64 * n));
    C25519_RC(t.reserve_dev(L, 1, 64 * n));
    C25519_TRY(hipMemcpyAsync(t.dbuf[L][0], priv, 64 * n, hipMemcpyHostToDevice, st));
    C25519_RC(ed25519_SignMessage_ragged_dev(t.dbuf[L][1], t.dbuf[L][0], d_msgs, (const uint64_t*)d_off, n, st));
    C25519_TRY(hipMemcpyAsync(sig, t.dbuf[L][1], 64 * n, hipMemcpyDeviceToHost, st));
    C25519_TRY(hipStreamSynchronize(st));
    return 0;
}

int ed25519_VerifySignature_ragged_batch(int* verdict, const unsigned char* sig, const unsigned char* pk,
                                         const unsigned char* msgs, const uint64_t* offsets, size_t n)
{
    C25519_API_CALL();
    if (!verdict || !sig || !pk || !offsets) return bad_arg("null pointer");
    if (n == 0) return 0;
    ThreadState& t = tls();
    C25519_RC(t.ensure());
    const int L = ThreadState::LANES - 1;
    hipStream_t st = t.stream[L];
    void *d_msgs, *d_off;
    C25519_RC(ragged_upload(t, &d_msgs, &d_off, msgs, offsets, n));
    C25519_RC(t.reserve_dev(L, 0, 64 * n));
    C25519_RC(t.reserve_dev(L, 1, 32 * n));
    C25519_RC(t.reserve_dev(L, 2, sizeof(int) * n));
    C25519_TRY(hipMemcpyAsync(t.dbuf[L][0], sig, 64 * n, hipMemcpyHostToDevice, st));
    C25519_TRY(hipMemcpyAsync(t.dbuf[L][1], pk, 32 * n, hipMemcpyHostToDevice, st));
    C25519_RC(ed25519_VerifySignature_ragged_dev(t.dbuf[L][2], t.dbuf[L][0], t.dbuf[L][1], d_msgs, (const uint64_t*)d_off, n, st));
    C25519_TRY(hipMemcpyAsync(verdict, t.dbuf[L][2], sizeof(int) * n, hipMemcpyDeviceToHost, st));
    C25519_TRY(hipStreamSynchronize(st));
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
    if (int rc = keypair_batch(pubKey, privKey, blinding, sk, 1)) c25519_host::die(__func__, rc);
}

void ed25519_SignMessage(unsigned char* signature, const unsigned char* privKey, const void* blinding,
                         const unsigned char* msg, size_t msg_size)
{
    if (int rc = sign_batch(signature, privKey, blinding, msg, msg_size, 1)) c25519_host::die(__func__, rc);
}

int ed25519_VerifySignature(const unsigned char* signature, const unsigned char* publicKey, const unsigned char* msg,
                            size_t msg_size)
{
    int verdict = 0;
    if (int rc = ed25519_VerifySignature_batch(&verdict, signature, publicKey, msg, msg_size, 1))
        c25519_host::die(__func__, rc);
    return verdict;
}

// Blinding contexts (ed25519_sign.c:289-341): 192 bytes, the reference's EDP_BLINDING_CTX shape (bl, zr, BP), derived
// ON THE DEVICE from the caller's seed; the context lives in the caller's storage or is malloc'ed here, exactly as in
// the reference.  Signing / key generation with a context computes (k + bl)*B + BP from a randomised starting point
// (lanes.cuh), so the walk and its table lookups see a scalar that differs per context; outputs are unchanged.
void* ed25519_Blinding_Init(void* context, const unsigned char* seed, size_t size)
{
    C25519_API_CALL_OR(nullptr);
    void* ctx = context ? context : malloc(4 * BLIND_WORDS);
    if (!ctx) return nullptr;                  // allocation failure is the only error the reference reports (:306)
    ThreadState& t = tls();
    auto run = [&]() -> int {
        C25519_RC(t.ensure());
        const int L = ThreadState::LANES - 1;
        C25519_RC(t.reserve_dev(L, 0, 4 * BLIND_WORDS));
        C25519_RC(t.reserve_dev(L, 1, size ? size : 1));
        if (size) C25519_TRY(hipMemcpyAsync(t.dbuf[L][1], seed, size, hipMemcpyHostToDevice, t.stream[L]));
        C25519_RC(ed25519_Blinding_Init_dev(t.dbuf[L][0], t.dbuf[L][1], size, t.stream[L]));
        C25519_TRY(hipMemcpyAsync(ctx, t.dbuf[L][0], 4 * BLIND_WORDS, hipMemcpyDeviceToHost, t.stream[L]));
        C25519_TRY(hipMemsetAsync(t.dbuf[L][0], 0, 4 * BLIND_WORDS, t.stream[L]));
        if (size) C25519_TRY(hipMemsetAsync(t.dbuf[L][1], 0, size, t.stream[L]));
        C25519_TRY(hipStreamSynchronize(t.stream[L]));
        return 0;
    };
    if (int rc = run()) c25519_host::die(__func__, rc);
    return ctx;
}

void ed25519_Blinding_Finish(void* context)
{
    if (context) {
        memset(context, 0, 4 * BLIND_WORDS);
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
    return run_batch(n, { Arr{ pk, nullptr, 32 }, Arr{ nullptr, ctx, 2080 } },
                     [&](void** d, size_t c, size_t, hipStream_t st) -> int {
                         return ed25519_Verify_Init_dev(d[1], d[0], c, st);
                     });
}

int ed25519_Verify_Check_batch(int* verdict, const void* ctx, const unsigned char* sig, const unsigned char* msg,
                               size_t msg_size, size_t n)
{
    C25519_API_CALL();
    if (!verdict || !ctx || !sig || (!msg && msg_size)) return bad_arg("null pointer");
    if (n == 0) return 0;
    ThreadState& t = tls();
    C25519_RC(t.ensure());
    // the reference's two-phase use is one Verify_Init and MANY Verify_Check calls on the same context
    // (ed25519_verify.c:282-286): the context has a device buffer of its own per calling thread and is uploaded only when
    // its bytes differ from what the thread uploaded last (a 2080-byte memcmp against a synchronous ~12 us copy per call)
    if (!t.vctx) C25519_TRY(hipMalloc(&t.vctx, 2080));
    void* dctx = t.vctx;
    if (!t.vctx_valid || memcmp(t.vctx_host, ctx, 2080) != 0) {
        t.vctx_valid = false;
        C25519_RC(c25519_host::upload_now(dctx, ctx, 2080));
        memcpy(t.vctx_host, ctx, 2080);
        t.vctx_valid = true;
    }
    return run_batch(n, { Arr{ sig, nullptr, 64 }, Arr{ msg, nullptr, msg_size }, Arr{ nullptr, verdict, sizeof(int) } },
                     [&](void** d, size_t c, size_t, hipStream_t st) -> int {
                         return ed25519_Verify_Check_dev(d[2], dctx, d[0], d[1], msg_size, c, st);
                     });
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
