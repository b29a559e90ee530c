// curve25519_amd/csrc/engine_common.cuh -- what the engine's translation units share: the device headers, the scratch layout
// between the kernels of a pass, the completion word, the shared inversion (k_batch_invert, its output encoders, launch_invert)
// and the host-side helpers of the *_dev entry points (tables, argument checks, launch policy).
//
// The engine is FOUR translation units, compiled in parallel by curve25519_amd/build.py and linked into one library:
//   engine_x25519.hip      the Montgomery-ladder kernels; curve25519_dh_CreateSharedKey / _CalculatePublicKey (*_dev)
//   engine_fixed_base.hip  the constant tables; key pairs, signatures, CalculatePublicKey_fast, blinding contexts (*_dev)
//   engine_verify.hip      verification: lattice path, reference order, two-phase / one key (*_dev)
//   engine_api.hip         library state, unit-test hooks, the host-pointer *_batch forms, the reference's single-call prototypes
// engine.hip includes all four as ONE translation unit: what the ISA tools, tests/test_resources.py and tools/build_variants.sh
// compile (the same kernels).  Entry points are declared in include/curve25519_amd.h, include/curve25519_dh.h and
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
#pragma once
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

// lanes per workgroup of the Ed25519 batch kernels (and waves per SIMD their register allocation aims at)
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

// a projective Edwards point into the scratch between the kernels of a pass
C25519_DEV void store_proj(const ProjScratch& scr, size_t n, size_t i, const ge_ext& S)
{
    soa_store_fe(scr.a, n, i, S.X);
    soa_store_fe(scr.b, n, i, S.Y);
    soa_store_fe(scr.z, n, i, S.Z);
}

constexpr int WB_BLOCK = 256;             // lanes per workgroup of the kernels that walk the wide comb (their parked column numbers: 10 KiB of LDS)

// (DoneWord / signal_done, the completion word of a call of ONE element: valu_gfx950.cuh)

// the records a one-element call carried in its kernel arguments (lanes.cuh: CallWords), laid down in LDS: the per-wave code then
// loads them through the pointers it is given, as it loads everything else (flat addresses).  One-wave workgroups only.
C25519_DEV const void* stage_call_words(u32* inl, const CallWords& cw)
{
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < CALL_WORDS; i++) inl[i] = cw.w[i];
    }
    __syncthreads();
    return inl;
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
    C25519_DEV bool skip() const { return false; }
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
    C25519_DEV bool skip() const { return false; }
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
    const u32* done_elsewhere;               // null, or a device word: non-zero = another kernel of the call has written the verdicts
    C25519_DEV bool skip() const { return done_elsewhere && *done_elsewhere; }
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

#ifndef C25519_INV_QUAD
#define C25519_INV_QUAD 1            // A/B switch: 0 = every lane of k_batch_invert inverts its own product (profiles/r06_ab_inv_quad.txt)
#endif
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
    if (fin.skip()) return;                                 // (uniform: a word of the call's scratch)
    constexpr bool PREFIX_IN_LDS = K > 14;
    __shared__ u32 pre_lds[PREFIX_IN_LDS ? (K - 1) * 10 * INV_BLOCK : 1];
    const size_t j = (size_t)blockIdx.x * INV_BLOCK + threadIdx.x;
    const bool live = j < m;                                // (a lane past the end stays: its quad shares the inversion below)
    fe z[K], pre[PREFIX_IN_LDS ? 1 : K];
    u32 zero_mask = 0;
#pragma unroll
    for (int t = 0; t < K; t++) {
        const size_t e = j + (size_t)t * m;
        if (live && e < n) soa_load_fe(z[t], Z, n, e);
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
    // ONE inversion per QUAD of lanes (4 K elements): the pairs' products, the quad's product T, 1 / T by the four lanes together
    // (fe_invert_quad: the division steps' three pairs on three lanes, ~7 700 instructions instead of one lane's ~13 700), then
    // each lane's own 1 / acc = (1 / T) * (the other pair's product) * (its partner's product)
    fe inv;
#if C25519_INV_QUAD
    {
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
        if (live && e < n) fin.emit(e, zi);
    }
}

// ================================================================================================
// host side, shared by the translation units
// ================================================================================================
namespace c25519_engine {

using c25519_host::Arr;
using c25519_host::ThreadState;
using c25519_host::aligned16;
using c25519_host::round_up;
using c25519_host::run_batch;
using c25519_host::bad_arg;
using c25519_host::tls;

// the constant tables of the current device, generated once per device per process (engine_fixed_base.hip)
int base_tables(const u32** limbs, const u32** bytes);
int wide_tables(const u32** wide);
// (engine_api.hip)
DoneWord take_done_word(size_t n);
CallWords call_words(size_t n, const void* rec0, const void* rec1);
CallWords call_record_and_message(size_t n, const void* rec, size_t rec_bytes, const void* msg, size_t msg_bytes);
int check_dev_args(size_t n, std::initializer_list<const void*> ptrs);
// (engine_fixed_base.hip) blinding: null or a device-resident 192-byte context
int keypair_dev(void* pub, void* priv, const void* sk, const void* blinding, size_t n, hipStream_t stream);
int sign_dev(void* sig, const void* priv, const void* blinding, Msgs msgs, size_t n, hipStream_t stream);

inline unsigned grid_for(size_t n, int block) { return (unsigned)((n + block - 1) / block); }
// the wide comb is the default: sign 824 against 643 M/s, key pairs 1110 against 815 M/s at 2^20 (profiles/r05_ab_base_comb.txt)
inline bool base_comb_wide() { return c25519_host::tunable_or(c25519_host::T_BASE_COMB, 1) == 1; }

// words of the projective-result part of the scratch for n elements (a, b, z, prefix; 16-byte aligned parts)
inline size_t proj_words(size_t n) { return 4 * round_up(SCR_FE * n, 4); }

inline ProjScratch carve_proj(u32* base, size_t n)
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

// a call of a few elements -- the reference's single-call prototypes are a batch of one -- runs ONE operation per wave
// (k_x25519_coop): ~5 x less latency than one operation per lane, at ~12 x the instructions per operation, so only while
// the waves still find idle SIMDs.  Tunable COOP_MAX = the largest such batch (A/B and test knob; 0 = never; at most 2^20:
// one workgroup per element).
inline bool coop_for(size_t n, size_t dflt)
{
    const long v = c25519_host::tunable(c25519_host::T_COOP_MAX);
    const size_t max = v == c25519_host::T_UNSET ? dflt : (size_t)std::min<long>(std::max<long>(v, 0), 1L << 20);
    return n <= max && c25519_host::batch_shape_hint() <= max;
}
// crossovers measured on MI355X (tools/small_batch_sweep.py, profiles/r04_small_batch_sweep.txt): the ladder one per wave
// wins up to 4096 elements (0.49 against 0.66 ms), the fixed-base operations up to 2048 (0.10-0.16 against 0.15-0.19 ms),
// verification (three waves per element, profiles/r05_small_batch_sweep.txt) up to 2048
inline bool x25519_coop_for(size_t n) { return coop_for(n, 4096); }
// ... two waves per element while every wave still finds a SIMD of its own: 167 against 179 us for one element, 193 against 201 for
// 512, 213 against 216 for 1024 (profiles/r05_small_batch_sweep.txt; tunable LADDER2_MAX; a per-wave call in any case)
inline bool x25519_two_waves_for(size_t n)
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
inline bool quad_for(size_t n, size_t dflt_min, size_t dflt_max)
{
    const long lo = c25519_host::tunable(c25519_host::T_QUAD_MIN), hi = c25519_host::tunable(c25519_host::T_QUAD_MAX);
    const size_t mn = lo == c25519_host::T_UNSET ? dflt_min : (size_t)std::max<long>(lo, 0);
    const size_t mx = hi == c25519_host::T_UNSET ? dflt_max : (size_t)std::min<long>(std::max<long>(hi, 0), 1L << 24);
    const size_t m = std::max(n, c25519_host::batch_shape_hint());     // a piece of a pipelined *_batch call: the whole call counts
    return m > mn && m <= mx;
}
inline bool x25519_quad_for(size_t n) { return quad_for(n, 3584, (size_t)1 << 15); }
inline bool verify_quad_for(size_t n) { return quad_for(n, 1024, (size_t)1 << 15); }      // k_ed25519_verify_quad_prep + _quad_walk: 0.30-0.31 ms up to 2^14, 0.48 at 2^15 (one-lane kernels: 0.52-0.63)
// the fixed-base operations on quads (k_ed25519_*_quad; over the wide comb, without a blinding context): one chain of 53-89 us up to
// 2^14 elements (one quad-wave per SIMD) against 81 us for 1024 per-wave signatures and the one-lane path's three launches
// (134-144 us at 2^15 / 2^16); profiles/r06_mid_batch_sweep.txt
inline bool fixed_base_quad_for(size_t n) { return quad_for(n, 1024, (size_t)1 << 14); }
// the one-key check over two wide combs on quads: 0.09-0.10 ms up to 2^14 pairs against the one-lane kernel's 0.16-0.18 (profiles/r06_one_key_rate.txt)
inline bool one_key_quad_for(size_t n) { return quad_for(n, 1024, (size_t)1 << 14); }
inline bool fixed_base_coop_for(size_t n) { return coop_for(n, 2048); }
inline bool verify_coop_for(size_t n) { return coop_for(n, 2048); }       // three waves per element: 0.13-0.55 against 0.60 ms (1.02 at 4096)

}  // namespace c25519_engine

using namespace c25519_engine;
