// curve25519_amd/csrc/engine_verify.hip -- Ed25519 verification: the lattice path (scalars, points, walk; on quads; one launch of three waves), the
// reference order, the two-phase calls with one key (shared table, two wide combs) -- kernels and *_dev entry points
// (one of the engine's four translation units: engine_common.cuh says which is which)
#include "engine_common.cuh"

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

// ... and on FOUR lanes per pair (quad::verify_check_wide_element: an addition in two product levels, inversion, encoding and the
// comparison in the same launch) for calls of 2^10 .. 2^14 pairs -- where the one-lane kernel above leaves three quarters of the
// SIMDs idle and every lane walks the whole 0.16 ms chain: what a caller with ONE remembered key and a few thousand signatures
// per call runs (ed25519_verify.c:282-286).  16 pairs per one-wave workgroup; LDS: the lanes' parked columns of s and h.
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 2)))
k_ed25519_verify_check_wide_quad(int* verdict, const void* sig, const u32* __restrict__ ctx, Msgs msgs, size_t n,
                                 const u32* __restrict__ wide_base, const u32* __restrict__ wide_key, const u32* __restrict__ wide_ok)
{
    if (!*wide_ok) return;                                 // k_ed25519_verify_check_shared decides this batch
    __shared__ unsigned short cols[2 * WB_COLS * 64];
    const size_t e = (size_t)blockIdx.x * quad::ELEMS_PER_WAVE + (threadIdx.x >> 2);
    if (e >= n) return;                                    // (whole quads leave)
    quad::verify_check_wide_element(verdict, sig, ctx, msgs.ptr(e), msgs.len(e), e, wide_base, wide_key, cols + threadIdx.x,
                                    cols + WB_COLS * 64 + threadIdx.x, 64);
}

namespace {

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

}  // namespace

extern "C" {

size_t ed25519_VerifySignature_scratch_bytes(size_t n) { return verify_scratch_bytes(n); }

static int verify_dev(void* verdict, const void* sig, const void* pk, Msgs msgs, size_t n, hipStream_t stream)
{
    // tunable VERIFY_REFERENCE_ORDER = 1: every element through the reference-order kernels -- Verify_Init's 4-fold table per
    // key, then the 4-fold + 8-fold walk of ed25519_verify.c:243-280: BASELINE.json configs[3] as worded (A/B and test knob)
    const bool fast = c25519_host::tunable_or(c25519_host::T_VERIFY_REFERENCE_ORDER, 0) == 0;
    if (int rc = check_dev_args(n, { verdict, sig, pk })) return rc;
    if (n == 0) return 0;
    return verify_run(sig, pk, msgs, n, stream, (int*)verdict, fast,
                      [&](const ProjScratch& scr) { return FinishVerify{ scr.a, scr.b, sig, (int*)verdict, n, nullptr }; });
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
    bool quads = false;
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
        quads = one_key_quad_for(n);
        if (quads)                                          // four lanes per pair, the verdict in the same launch
            k_ed25519_verify_check_wide_quad<<<grid_for(n, quad::ELEMS_PER_WAVE), 64, 0, stream>>>(
                (int*)verdict, sig, (const u32*)ctx, Msgs{ (const uint8_t*)msg, msg_size, nullptr }, n, wide_base, wide_key, wide_ok);
        else
            k_ed25519_verify_check_wide<<<grid_for(n, WB_BLOCK), WB_BLOCK, 0, stream>>>(
                scr, sig, (const u32*)ctx, Msgs{ (const uint8_t*)msg, msg_size, nullptr }, n, wide_base, wide_key, wide_ok);
        C25519_TRY(hipGetLastError());
    }
    k_ed25519_verify_check_shared<<<grid_for(n, ED_BLOCK), ED_BLOCK, 0, stream>>>(
        scr, sig, (const u32*)ctx, Msgs{ (const uint8_t*)msg, msg_size, nullptr }, n, tbl, wide_ok);
    C25519_TRY(hipGetLastError());
    // (the quad kernel has written the verdicts itself where the combs decided: the shared inversion then finds wide_ok set and leaves)
    C25519_RC(launch_invert(scr, n, FinishVerify{ scr.a, scr.b, sig, (int*)verdict, n, quads ? wide_ok : nullptr }, stream));
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

}  // extern "C"
