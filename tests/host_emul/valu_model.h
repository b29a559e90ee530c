// tests/host_emul/valu_model.h -- TEST INFRASTRUCTURE.  A C model of the gfx950 primitives of
// curve25519_amd/csrc/valu_gfx950.cuh plus stand-ins for the HIP keywords the device headers use, so that the
// SAME device source (fe25519.cuh, sc25519.cuh, sha512.cuh, ge25519.cuh, x25519.cuh, lanes.cuh) can be compiled by
// g++ and unit-tested on the CPU, one "lane" at a time, against Python big integers and the committed fixtures.
// It is force-included (-include) by tests/host_emul/build.py only; nothing in the product links or includes it,
// and it is not a fallback: libcurve25519_amd.so has no host arithmetic at all.
#pragma once
#define C25519_VALU_PRIMITIVES 1
#include <pthread.h>
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#define __device__
#define __forceinline__ inline
#define __restrict__
#define C25519_DEV inline
#define C25519_SCHED_FENCE() ((void)0)
#define C25519_VOP2_RUN_BEGIN() ((void)0)       // wave priority around runs of VOP2 instructions: nothing to model
#define C25519_VOP2_RUN_END() ((void)0)

struct uint4 { uint32_t x, y, z, w; };
static inline uint4 make_uint4(uint32_t x, uint32_t y, uint32_t z, uint32_t w) { return uint4{ x, y, z, w }; }
struct emul_dim3 { unsigned x, y, z; };
static const emul_dim3 blockIdx = { 0, 0, 0 }, blockDim = { 1, 1, 1 };
// threadIdx: 0 for the one-lane-at-a-time tests; the lane number while coop_wave.h runs a workgroup of lock-step lanes
extern thread_local emul_dim3 emul_tid;
#define threadIdx emul_tid
#include "coop_wave.h"
static inline void __syncthreads() { emul_coop::block_sync(); }
// the cross-lane instructions of coop25519.cuh, as rendezvous between the lanes coop_wave.h runs
#define __builtin_amdgcn_update_dpp(old, src, ctrl, row_mask, bank_mask, bound_ctrl) \
    ((int)emul_coop::dpp((uint32_t)(old), (uint32_t)(src), (ctrl), (bound_ctrl)))
#define __builtin_amdgcn_permlane16_swap(a, b, fi, bc) emul_coop::swap16((a), (b))
#define __builtin_amdgcn_permlane32_swap(a, b, fi, bc) emul_coop::swap32((a), (b))
#define __builtin_amdgcn_wave_barrier() emul_coop::wave_sync()
static inline uint32_t atomicAdd(uint32_t* p, uint32_t v) { const uint32_t old = *p; *p = old + v; return old; }   // (lanes run one at a time)
static inline int __syncthreads_or(int p) { return p; }
// __any(): a "wave" of one lane by default.  emul_wave_run() (emul.cpp) runs G host threads as the lanes of one wave in
// lock-step: each lane's code then meets every __any at the same point (the device source only branches on wave-uniform
// values around them) and gets the OR over the group -- so the divergent paths of the lattice reduction (lanes idling
// while others still iterate, the word-shift hints, loops that end when the LAST lane is done) run on the CPU too.
struct EmulWave {
    int lanes;
    int arrived, generation;
    bool acc, result;
    pthread_mutex_t mu;
    pthread_cond_t cv;
};
extern thread_local EmulWave* emul_wave;                 // the wave this host thread is a lane of (nullptr: alone)
static inline bool __any(bool p)
{
    EmulWave* w = emul_wave;
    if (!w) return p;
    pthread_mutex_lock(&w->mu);
    w->acc = w->acc || p;
    const int gen = w->generation;
    if (++w->arrived == w->lanes) {                       // last one in: publish and release the others
        w->result = w->acc;
        w->acc = false;
        w->arrived = 0;
        w->generation++;
        pthread_cond_broadcast(&w->cv);
    } else {
        while (w->generation == gen) pthread_cond_wait(&w->cv, &w->mu);
    }
    const bool r = w->result;
    pthread_mutex_unlock(&w->mu);
    return r;
}

namespace c25519 {

typedef uint32_t u32;
typedef uint64_t u64;

// loop-trip counters of the lattice reduction (verify_fast.cuh), read by tests/test_verify_fast.py
struct LatCounters { unsigned long long lehmer_outer, lehmer_inner, exact_steps; };
extern LatCounters emul_lat_counters;
#define C25519_LAT_COUNT(what) __atomic_fetch_add(&::c25519::emul_lat_counters.what, 1ULL, __ATOMIC_RELAXED)

inline u32 dbl32(u32 x) { return x + x; }
inline float fast_div(float a, float b) { return a / b; }
inline u32 xor3_32(u32 a, u32 b, u32 c) { return a ^ b ^ c; }                                  // v_bitop3_b32 0x96
inline u32 ch_32(u32 e, u32 f, u32 g) { return (e & f) | (~e & g); }                           // 0xca
inline u32 maj_32(u32 a, u32 b, u32 c) { return (a & b) | (a & c) | (b & c); }                 // 0xe8
struct DoneWord { u32* word; u32 seq; };                                                      // (valu_gfx950.cuh: the completion word)
inline void signal_done(const DoneWord& d) { if (d.word) __atomic_store_n(d.word, d.seq, __ATOMIC_RELEASE); }
// valu_gfx950.cuh: row_carry, the same moves in the same order (every lane of the wave meets them together)
inline u32 row_carry(u64 S, u32 w, u32 mask, u32 mask_next, u32 m1, u32 m2)
{
    auto mv = [](u32 x, int ctrl) { return (u32)emul_coop::dpp(0u, x, ctrl, true); };
    const u32 l0 = (u32)S & mask, l1 = (u32)(S >> w) & mask_next, l2 = (u32)(S >> 51);
    u32 a = mv(l2, 0x112), b = mv(l2, 0x128);              // row_shr:2, row_ror:8
    u32 limb = l0 + a + b * m2;
    a = mv(l1, 0x111); b = mv(l1, 0x127);                  // row_shr:1, row_ror:7
    limb += a + b * m1;
    const u32 e = limb >> w;
    limb &= mask;
    a = mv(e, 0x111); b = mv(e, 0x127);
    return limb + a + b * m1;
}
inline u64 pair64(u32 lo, u32 hi) { return ((u64)hi << 32) | lo; }
inline u32 alignbit32(u32 hi, u32 lo, int s) { return (u32)((((u64)hi << 32) | lo) >> s); }

// v_mad_u64_u32 wraps modulo 2^64 exactly like unsigned C arithmetic; the model additionally REPORTS a wrap,
// because the field layer's bound contract says it can never happen (tools/fe_bounds.py proves it; this checks it
// on the values that actually flow through the tests).
extern unsigned long long emul_mad_overflows;
// ... and COUNTS the instruction: every v_mad_u64_u32 the device source issues goes through here, so a run of one
// operation on the model is a count of the MADs the kernels execute for it (tools/executed_macs.py -> bench.py's
// roofline.valu.executed_macs_per_op)
extern unsigned long long emul_mad_count;
#define C25519_COUNT_MAD(n) __atomic_fetch_add(&::c25519::emul_mad_count, (unsigned long long)(n), __ATOMIC_RELAXED)
inline u64 mad64(u64 acc, u32 x, u32 y)
{
    __atomic_fetch_add(&emul_mad_count, 1ULL, __ATOMIC_RELAXED);
    const u64 p = (u64)x * y;
    const u64 r = acc + p;
    if (r < acc) emul_mad_overflows++;
    return r;
}
inline int64_t mad_i64_i32(int64_t acc, int32_t x, int32_t y)      // v_mad_i64_i32; a signed wrap would break safegcd25519.cuh's bounds
{
    __atomic_fetch_add(&emul_mad_count, 1ULL, __ATOMIC_RELAXED);
    int64_t r;
    if (__builtin_add_overflow(acc, (int64_t)x * y, &r)) emul_mad_overflows++;
    return r;
}
inline int64_t mad2_i64_i32(int64_t acc, int32_t x0, int32_t y0, int32_t x1, int32_t y1) { return mad_i64_i32(mad_i64_i32(acc, x0, y0), x1, y1); }
// thirty division steps on low words (valu_gfx950.cuh: sg_steps30 / sg_steps30_quad), instruction by instruction
inline void sg_model_pair(u32& X, u32& Y, u32 c1, u32 c2, u32 m, u32 n)
{
    const u32 t = (X ^ c1) & c2;                  // v_bitop3_b32 0x28
    Y = Y + t + n;                                // v_add3_u32
    X = (X + (Y & m)) << 1;                       // v_and_b32, v_add_lshl_u32
}
inline void sg_steps30(int32_t& zeta, u32& f, u32& g, u32& u, u32& q, u32& v, u32& r)
{
    for (int i = 0; i < 30; i++) {
        const u32 c2 = (u32)0 - ((g >> i) & 1u), c1 = (u32)(zeta >> 31), m = c1 & c2, n = m >> 31;
        sg_model_pair(f, g, c1, c2, m, n);
        sg_model_pair(u, q, c1, c2, m, n);
        sg_model_pair(v, r, c1, c2, m, n);
        zeta = (int32_t)(((u32)zeta ^ m) - 1u);   // v_xad_u32
    }
}
inline void sg_steps30_quad(int32_t& zeta, u32& X, u32& Y)
{
    for (int i = 0; i < 30; i++) {
        const u32 g0 = (u32)__builtin_amdgcn_update_dpp(0, (int)Y, 0, 0xf, 0xf, true);     // quad_perm:[0,0,0,0]
        const u32 c2 = (u32)0 - ((g0 >> i) & 1u), c1 = (u32)(zeta >> 31), m = c1 & c2, n = m >> 31;
        sg_model_pair(X, Y, c1, c2, m, n);
        zeta = (int32_t)(((u32)zeta ^ m) - 1u);
    }
}
inline u64 mad_chain5(u64 acc, const u32 (&x)[5], const u32 (&y)[5])
{
    for (int t = 0; t < 5; t++) acc = mad64(acc, x[t], y[t]);
    return acc;
}
inline u64 mad_chain6(u64 acc, const u32 (&x)[6], const u32 (&y)[6])
{
    for (int t = 0; t < 6; t++) acc = mad64(acc, x[t], y[t]);
    return acc;
}
inline u64 mad_chain5_from_zero(const u32 (&x)[5], const u32 (&y)[5]) { return mad_chain5(0, x, y); }
inline u64 mad_chain6_from_zero(const u32 (&x)[6], const u32 (&y)[6]) { return mad_chain6(0, x, y); }
inline u64 mad_chain10(u64 acc, const u32 (&x)[10], const u32 (&y)[10]);
inline u64 mad_chain10_from_zero(const u32 (&x)[10], const u32 (&y)[10]) { return mad_chain10(0, x, y); }
inline u64 mad_chain10(u64 acc, const u32 (&x)[10], const u32 (&y)[10])
{
    for (int t = 0; t < 10; t++) acc = mad64(acc, x[t], y[t]);
    return acc;
}

}  // namespace c25519
