// curve25519_amd/csrc/engine_x25519.hip -- the X25519 kernels (the Montgomery ladder per lane, fused with the shared inversion, per wave, on two
// waves, on quads) and curve25519_dh_CreateSharedKey_dev / curve25519_dh_CalculatePublicKey_dev
// (one of the engine's four translation units: engine_common.cuh says which is which)
#include "engine_common.cuh"

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
#if C25519_INV_QUAD
        {   // one inversion per quad of the wave's lanes (k_batch_invert's exchange, engine_common.cuh)
            fe partner, pair, other_pair, total;
            quad::fe_qperm<1, 0, 3, 2>(partner, acc);
            fe_mul(pair, acc, partner);
            quad::fe_qperm<2, 3, 0, 1>(other_pair, pair);
            fe_mul(total, pair, other_pair);
            fe_invert_quad(inv, total);
            fe_mul(inv, inv, other_pair);
            fe_mul(inv, inv, partner);
        }
#else
        fe_invert(inv, acc);
#endif
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

// One operation per WAVE (coop25519.cuh): what a call of a few elements runs -- the reference's own single-call
// prototypes above all.  Ladder, doublings, inversion and the last multiplication are cooperative (a field element
// limb-per-lane, up to four products at a time); only the decoding of the inputs and the canonical encoding of the result
// are the batch kernels' per-lane code, run by every lane on the same values.
template <bool BASE9>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 4))) k_x25519_coop(void* out, const void* pk, void* sk, size_t n, DoneWord done, CallWords cw)
{
    __shared__ __attribute__((aligned(16))) u32 lds[coop::ROWQ_OFF];
    if (blockIdx.x >= n) return;
    coop::x25519_one<BASE9>(lds, coop::make_lane(threadIdx.x), out, pk, sk, blockIdx.x, &cw, &done);    // (signals behind the result, before its LDS wipe)
}

// ... and on TWO waves per element (coop::x25519_two_waves: a ladder step in two product levels -- the differential addition with
// x1 times the sum carried along on one wave, the doubling on the other, one workgroup barrier per step): what ONE
// curve25519_dh_CreateSharedKey call and calls of up to 512 run -- 183 -> 168 us per call
__global__ void __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(1, 4))) k_x25519_coop2(void* out, const void* pk, void* sk, size_t n, DoneWord done, CallWords cw)
{
    __shared__ __attribute__((aligned(16))) u32 lds[coop::X2_LDS_WORDS];
    if (blockIdx.x >= n) return;
    coop::x25519_two_waves(lds, out, pk, sk, blockIdx.x, &cw, &done);    // (wave 0 stores, signals, wipes; wave 1 has left inside)
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

namespace {

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

#ifdef C25519_CYCLE_PROBE
// measurement builds only: where the waves of k_x25519_fused write their stamps (PROBE_WORDS u64 per wave), or null
int c25519_amd_probe_set(void* buf)
{
    C25519_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_cycle_probe), &buf, sizeof buf));
    return 0;
}
int c25519_amd_probe_words(void) { return PROBE_WORDS; }
#endif

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

}  // extern "C"
