// tools/ubench/valu_rates.hip -- measures the sustained issue rate of the VALU instructions the
// field arithmetic is built from, on the device it runs on (gfx950).  The v_mad_u64_u32 figure is the
// denominator of the integer-MAC roofline reported by bench.py ("measured device
// integer-multiply peak").  Build: hipcc --offload-arch=gfx950 -O3 valu_rates.hip -o valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
    fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

// A loop trip issues INNER x UNROLL instructions of the tested kind.  The trip has to be LONG: with 16 per trip the loop
// overhead (and where the loop top falls in a 64-byte fetch line: tools/ubench/mad_peak.hip) costs 10-25 % -- that is
// how round 1 and the first half of round 2 came to quote 28-32 T lane-op/s for v_mad_u64_u32 and 57-67 T for the
// full-rate class, when the instructions themselves run at 37.7 T (plain half rate) and ~75 T.
constexpr int ITERS = 512;
constexpr int INNER = 8;
constexpr int UNROLL = 16;      // instructions of the tested kind per inner repetition

// Each body issues UNROLL instructions on 8 independent dependency chains (2 per chain).
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

// s_memtime ticks at the shader clock (MI355X_MICROARCH.md); wave 0 of workgroup 0 reports the ticks its own loop took,
// which gives the shader clock the chip actually sustained under this instruction mix (ticks / elapsed time) and
// cycles per wave-instruction that do not depend on the nominal 2.4 GHz.
template <int KIND>
__global__ void __launch_bounds__(256) k_rate(unsigned* out, unsigned seed, unsigned long long* ticks)
{
    const unsigned long long t_begin = __builtin_readcyclecounter();
    unsigned a = threadIdx.x * 2654435761u + seed, b = a ^ 0x9e3779b9u;
    unsigned r0 = a, r1 = b, r2 = a + 1, r3 = b + 1, r4 = a + 2, r5 = b + 2, r6 = a + 3, r7 = b + 3;
    unsigned t0 = b, t1 = a, t2 = b + 5, t3 = a + 5, t4 = b + 6, t5 = a + 6, t6 = b + 7, t7 = a + 7;
    unsigned long long q0 = a, q1 = b, q2 = r2, q3 = r3, q4 = r4, q5 = r5, q6 = r6, q7 = r7;
    double d0 = a, d1 = b, d2 = 1.0, d3 = 2.0, d4 = 3.0, d5 = 4.0, d6 = 5.0, d7 = 6.0;
    double dm = 1.0000001, da = 1e-9;
    float f0 = a, f1 = b, f2 = 1, f3 = 2, f4 = 3, f5 = 4, f6 = 5, f7 = 6, fm = 1.0001f, fa = 1e-5f;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
      for (int u = 0; u < INNER; ++u) {
        if constexpr (KIND == 0) {          // v_mad_u64_u32, accumulate chain
#define X(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(q##i) : "v"(a), "v"(b) : "vcc");
            REP8(X) REP8(X)
#undef X
        } else if constexpr (KIND == 1) {   // v_mul_lo_u32
#define X(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(r##i) : "v"(a));
            REP8(X) REP8(X)
#undef X
        } else if constexpr (KIND == 2) {   // v_mul_hi_u32
#define X(i) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(r##i) : "v"(a));
            REP8(X) REP8(X)
#undef X
        } else if constexpr (KIND == 3) {   // v_add_u32
#define X(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(r##i) : "v"(a));
            REP8(X) REP8(X)
#undef X
        } else if constexpr (KIND == 4) {   // v_add_co_u32 / v_addc_co_u32 pair (counts as 2)
#define X(i) asm volatile("v_add_co_u32 %0, vcc, %0, %2\n\tv_addc_co_u32 %1, vcc, %1, %3, vcc" : "+v"(r##i), "+v"(t##i) : "v"(a), "v"(b) : "vcc");
            REP8(X)
#undef X
        } else if constexpr (KIND == 5) {   // v_add3_u32
#define X(i) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(r##i) : "v"(a), "v"(b));
            REP8(X) REP8(X)
#undef X
        } else if constexpr (KIND == 6) {   // v_fma_f64
#define X(i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d##i) : "v"(dm), "v"(da));
            REP8(X) REP8(X)
#undef X
        } else if constexpr (KIND == 7) {   // v_mad_u32_u24
#define X(i) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(r##i) : "v"(a), "v"(b));
            REP8(X) REP8(X)
#undef X
        } else if constexpr (KIND == 8) {   // v_lshl_add_u64 (64-bit add)
#define X(i) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(q##i) : "v"(q7));
            REP8(X) REP8(X)
#undef X
        } else if constexpr (KIND == 9) {   // v_alignbit_b32
#define X(i) asm volatile("v_alignbit_b32 %0, %0, %1, 7" : "+v"(r##i) : "v"(a));
            REP8(X) REP8(X)
#undef X
        } else if constexpr (KIND == 10) {  // v_lshrrev_b64
#define X(i) asm volatile("v_lshrrev_b64 %0, 3, %0" : "+v"(q##i));
            REP8(X) REP8(X)
#undef X
        } else if constexpr (KIND == 11) {  // v_mul_hi_u32_u24
#define X(i) asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(r##i) : "v"(a));
            REP8(X) REP8(X)
#undef X
        } else if constexpr (KIND == 12) {  // v_fma_f32
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f##i) : "v"(fm), "v"(fa));
            REP8(X) REP8(X)
#undef X
        } else if constexpr (KIND == 13) {  // v_cndmask_b32 (vcc)
#define X(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r##i) : "v"(a) : );
            REP8(X) REP8(X)
#undef X
        } else if constexpr (KIND == 14) {  // v_xor_b32
#define X(i) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(r##i) : "v"(a));
            REP8(X) REP8(X)
#undef X
        } else if constexpr (KIND == 15) {  // mixed: 1 mad_u64_u32 + 1 addc per pair (counts 16 = 8 mad + 8 addc)
#define X(i) asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(q##i), "+v"(r##i) : "v"(a), "v"(b) : "vcc");
            REP8(X)
#undef X
        } else if constexpr (KIND == 16) {  // mixed: 1 mad_u64_u32 + 3 v_add_u32 (counts 16 = 4 mad + 12 add)
#define X(i) asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_add_u32 %1, %1, %2\n\tv_add_u32 %1, %1, %3\n\tv_add_u32 %1, %1, %2" : "+v"(q##i), "+v"(r##i) : "v"(a), "v"(b) : "vcc");
            X(0) X(1) X(2) X(3)
#undef X
        } else if constexpr (KIND == 17) {  // v_pk_fma_f32
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(d##i) : "v"(dm), "v"(da));
            REP8(X) REP8(X)
#undef X
        } else if constexpr (KIND == 18) {  // v_mul_u32_u24
#define X(i) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(r##i) : "v"(a));
            REP8(X) REP8(X)
#undef X
        } else if constexpr (KIND == 19) {  // v_mul_f64
#define X(i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d##i) : "v"(dm));
            REP8(X) REP8(X)
#undef X
        } else if constexpr (KIND == 20) {  // v_add_f64
#define X(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d##i) : "v"(da));
            REP8(X) REP8(X)
#undef X
        } else if constexpr (KIND == 21) {  // v_cvt_f64_u32 + v_cvt_u32_f64 pair (counts 2 each -> 16)
#define X(i) asm volatile("v_cvt_f64_u32 %0, %1\n\tv_cvt_u32_f64 %1, %0" : "+v"(d##i), "+v"(r##i));
            REP8(X)
#undef X
        } else if constexpr (KIND == 22) {  // mad_u64_u32 with SGPR-free null carry: v_mad_u64_u32 dst, s[..]
#define X(i) asm volatile("v_mad_u64_u32 %0, s[20:21], %1, %2, %0" : "+v"(q##i) : "v"(a), "v"(b) : "s20", "s21");
            REP8(X) REP8(X)
#undef X
        } else if constexpr (KIND == 24) {  // v_cndmask_b32 VOP3 form, mask in an SGPR pair
#define X(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[22:23]" : "+v"(r##i) : "v"(a) : );
            REP8(X) REP8(X)
#undef X
        } else if constexpr (KIND == 25) {  // v_bfi_b32 (bitwise select)
#define X(i) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(r##i) : "v"(a), "v"(b));
            REP8(X) REP8(X)
#undef X
        } else if constexpr (KIND == 26) {  // v_and_or_b32
#define X(i) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(r##i) : "v"(a), "v"(b));
            REP8(X) REP8(X)
#undef X
        } else if constexpr (KIND == 27) {  // v_cndmask_b32 with vcc freshly written by a v_cmp each time
#define X(i) asm volatile("v_cmp_lt_u32 vcc, %1, %0\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(r##i) : "v"(a) : "vcc");
            REP8(X)
#undef X
        } else if constexpr (KIND == 28) {  // v_lshlrev_b32 (VOP2 shift)
#define X(i) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(r##i));
            REP8(X) REP8(X)
#undef X
        } else if constexpr (KIND == 29) {  // v_sub_u32
#define X(i) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(r##i) : "v"(a));
            REP8(X) REP8(X)
#undef X
        } else if constexpr (KIND == 30) {  // v_and_b32
#define X(i) asm volatile("v_and_b32 %0, %0, %1" : "+v"(r##i) : "v"(a));
            REP8(X) REP8(X)
#undef X
        } else if constexpr (KIND == 31) {  // v_mov_b32
#define X(i) asm volatile("v_mov_b32 %0, %1" : "=v"(r##i) : "v"(a));
            REP8(X) REP8(X)
#undef X
        } else if constexpr (KIND == 32) {  // v_mad_u64_u32 with a zero addend (inline constant): no 64-bit VGPR source
#define X(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(q##i) : "v"(r##i), "v"(b) : "vcc");
            REP8(X) REP8(X)
#undef X
        } else if constexpr (KIND == 33) {  // v_mad_u64_u32, ONE dependent chain (latency): 16 MADs on one accumulator
#define X(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(q0) : "v"(a), "v"(b) : "vcc");
            REP8(X) REP8(X)
#undef X
        } else if constexpr (KIND == 34) {  // v_mad_u64_u32 with an SGPR multiplicand
#define X(i) asm volatile("v_mad_u64_u32 %0, vcc, s24, %1, %0" : "+v"(q##i) : "v"(b) : "vcc", "s24");
            REP8(X) REP8(X)
#undef X
        } else if constexpr (KIND == 35) {  // v_mad_i64_i32
#define X(i) asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(q##i) : "v"(a), "v"(b) : "vcc");
            REP8(X) REP8(X)
#undef X
        } else if constexpr (KIND == 36) {  // column chain as the field product issues it: 10 MADs + 64-bit shift + mask
#define M2 "v_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %2, %1, %0\n\t"
            asm volatile(M2 M2 M2 M2 "v_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %2, %1, %0"
                         : "+v"(q0) : "v"(a), "v"(b) : "vcc");
            r0 ^= (unsigned)q0 & 0x3ffffffu;
            q0 >>= 26;
            asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %2, %1, %0"
                         : "+v"(q0) : "v"(a), "v"(b) : "vcc");
            r1 ^= (unsigned)q0 & 0x1ffffffu;
            q0 >>= 25;
#undef M2
        } else if constexpr (KIND == 37) {  // v_fma_f32, 16 independent chains, destination != sources
#define X(i) asm volatile("v_fma_f32 %0, %2, %3, %1" : "=v"(f##i) : "v"(fa), "v"(fm), "v"(fa));
            REP8(X) REP8(X)
#undef X
        } else if constexpr (KIND == 38) {  // v_add_u32 with an inline constant (one VGPR read)
#define X(i) asm volatile("v_add_u32 %0, 17, %0" : "+v"(r##i));
            REP8(X) REP8(X)
#undef X
        } else if constexpr (KIND == 39) {  // v_pk_fma_f32 counted as TWO lane-FMAs per lane
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(d##i) : "v"(dm), "v"(da));
            REP8(X) REP8(X)
#undef X
        } else if constexpr (KIND == 40) {  // v_mad_u64_u32 + v_addc_co_u32 as a saturated (radix 2^32) product needs them
#define X(i) asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(q##i), "+v"(r##i) : "v"(a), "v"(b) : "vcc");
            REP8(X)
#undef X
        } else if constexpr (KIND == 41) {  // v_lshl_add_u32
#define X(i) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(r##i) : "v"(a));
            REP8(X) REP8(X)
#undef X
        } else if constexpr (KIND == 42) {  // v_or_b32
#define X(i) asm volatile("v_or_b32 %0, %0, %1" : "+v"(r##i) : "v"(a));
            REP8(X) REP8(X)
#undef X
        } else if constexpr (KIND == 43) {  // v_mad_u64_u32, distinct destination (d = a*b + c, c another pair)
#define X(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %3" : "=v"(q##i) : "v"(a), "v"(b), "v"(q7) : "vcc");
            X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(0) X(1)
#undef X
        } else if constexpr (KIND == 44) {  // dependent MAD chain, ping-pong between two accumulators (dst != src2)
#define X(i) asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %1\n\tv_mad_u64_u32 %1, vcc, %3, %2, %0" : "+v"(q0), "+v"(q1) : "v"(a), "v"(b) : "vcc");
            REP8(X)
#undef X
        } else if constexpr (KIND == 45) {  // MAD, explicit registers, all four source dwords in different VGPR banks (v%4)
            asm volatile(
                "v_mov_b32 v40, %0\n\tv_mov_b32 v41, %1\n\t"
                "v_mad_u64_u32 v[42:43], vcc, v40, v41, v[42:43]\n\tv_mad_u64_u32 v[46:47], vcc, v40, v41, v[46:47]\n\t"
                "v_mad_u64_u32 v[50:51], vcc, v40, v41, v[50:51]\n\tv_mad_u64_u32 v[54:55], vcc, v40, v41, v[54:55]\n\t"
                "v_mad_u64_u32 v[42:43], vcc, v40, v41, v[42:43]\n\tv_mad_u64_u32 v[46:47], vcc, v40, v41, v[46:47]\n\t"
                "v_mad_u64_u32 v[50:51], vcc, v40, v41, v[50:51]\n\tv_mad_u64_u32 v[54:55], vcc, v40, v41, v[54:55]\n\t"
                "v_mad_u64_u32 v[42:43], vcc, v40, v41, v[42:43]\n\tv_mad_u64_u32 v[46:47], vcc, v40, v41, v[46:47]\n\t"
                "v_mad_u64_u32 v[50:51], vcc, v40, v41, v[50:51]\n\tv_mad_u64_u32 v[54:55], vcc, v40, v41, v[54:55]\n\t"
                "v_mad_u64_u32 v[42:43], vcc, v40, v41, v[42:43]\n\tv_mad_u64_u32 v[46:47], vcc, v40, v41, v[46:47]\n\t"
                "v_mad_u64_u32 v[50:51], vcc, v40, v41, v[50:51]\n\tv_mad_u64_u32 v[54:55], vcc, v40, v41, v[54:55]\n\t"
                "v_xor_b32 %2, v42, v46\n\tv_xor_b32 %2, %2, v50\n\tv_xor_b32 %2, %2, v54"
                : : "v"(a), "v"(b), "v"(r0)
                : "vcc", "v40", "v41", "v42", "v43", "v46", "v47", "v50", "v51", "v54", "v55");
        } else if constexpr (KIND == 46) {  // MAD, explicit registers, both factors and the addend's low dword in ONE bank
            asm volatile(
                "v_mov_b32 v40, %0\n\tv_mov_b32 v44, %1\n\t"
                "v_mad_u64_u32 v[48:49], vcc, v40, v44, v[48:49]\n\tv_mad_u64_u32 v[52:53], vcc, v40, v44, v[52:53]\n\t"
                "v_mad_u64_u32 v[56:57], vcc, v40, v44, v[56:57]\n\tv_mad_u64_u32 v[60:61], vcc, v40, v44, v[60:61]\n\t"
                "v_mad_u64_u32 v[48:49], vcc, v40, v44, v[48:49]\n\tv_mad_u64_u32 v[52:53], vcc, v40, v44, v[52:53]\n\t"
                "v_mad_u64_u32 v[56:57], vcc, v40, v44, v[56:57]\n\tv_mad_u64_u32 v[60:61], vcc, v40, v44, v[60:61]\n\t"
                "v_mad_u64_u32 v[48:49], vcc, v40, v44, v[48:49]\n\tv_mad_u64_u32 v[52:53], vcc, v40, v44, v[52:53]\n\t"
                "v_mad_u64_u32 v[56:57], vcc, v40, v44, v[56:57]\n\tv_mad_u64_u32 v[60:61], vcc, v40, v44, v[60:61]\n\t"
                "v_mad_u64_u32 v[48:49], vcc, v40, v44, v[48:49]\n\tv_mad_u64_u32 v[52:53], vcc, v40, v44, v[52:53]\n\t"
                "v_mad_u64_u32 v[56:57], vcc, v40, v44, v[56:57]\n\tv_mad_u64_u32 v[60:61], vcc, v40, v44, v[60:61]\n\t"
                "v_xor_b32 %2, v48, v52\n\tv_xor_b32 %2, %2, v56\n\tv_xor_b32 %2, %2, v60"
                : : "v"(a), "v"(b), "v"(r0)
                : "vcc", "v40", "v44", "v48", "v49", "v52", "v53", "v56", "v57", "v60", "v61");
        } else if constexpr (KIND == 47) {  // v_mad_u64_u32 with src0 == src1 (squaring term: 3 source dwords)
#define X(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %1, %0" : "+v"(q##i) : "v"(a) : "vcc");
            REP8(X) REP8(X)
#undef X
        } else if constexpr (KIND == 48) {  // the ladder's select: ONE v_cmp writes vcc, 16 v_cndmask_b32 (VOP2) read it
            asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(a), "v"(b) : "vcc");
#define X(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r##i) : "v"(a) : );
            REP8(X) REP8(X)
#undef X
        } else if constexpr (KIND == 49) {  // select by mask arithmetic: r ^= (r ^ a) & m   (3 full-rate VOP2 per select; counts 15 + 1)
#define X(i) asm volatile("v_xor_b32 %1, %0, %2\n\tv_and_b32 %1, %1, %3\n\tv_xor_b32 %0, %0, %1" : "+v"(r##i), "+v"(t##i) : "v"(a), "v"(b));
            X(0) X(1) X(2) X(3) X(4)
            asm volatile("v_xor_b32 %0, %0, %1" : "+v"(r5) : "v"(a));
#undef X
        } else if constexpr (KIND == 50) {  // v_cndmask_b32 (VOP2, vcc) with vcc written by an SALU move every iteration
            asm volatile("s_mov_b64 vcc, s[22:23]" : : : "vcc");
#define X(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r##i) : "v"(a) : );
            REP8(X) REP8(X)
#undef X
        } else if constexpr (KIND == 51) {  // v_cndmask_b32(vcc) alternating with an independent v_xor_b32 (8 + 8)
            asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(a), "v"(b) : "vcc");
#define X(i) asm volatile("v_cndmask_b32 %0, %0, %2, vcc\n\tv_xor_b32 %1, %1, %2" : "+v"(r##i), "+v"(t##i) : "v"(a) : );
            REP8(X)
#undef X
        } else if constexpr (KIND == 52) {  // v_cndmask_b32(vcc) with 3 independent VALU ops between (4 + 12)
            asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(a), "v"(b) : "vcc");
#define X(i) asm volatile("v_cndmask_b32 %0, %0, %2, vcc\n\tv_xor_b32 %1, %1, %2\n\tv_add_u32 %1, %1, %2\n\tv_xor_b32 %1, %1, %0" : "+v"(r##i), "+v"(t##i) : "v"(a) : );
            X(0) X(1) X(2) X(3)
#undef X
        } else if constexpr (KIND == 23) {  // v_mad_i32_i24 ... placeholder for v_perm/v_bfe: v_bfe_u32
#define X(i) asm volatile("v_bfe_u32 %0, %0, 3, 29" : "+v"(r##i));
            REP8(X) REP8(X)
#undef X
        }
      }
    }
    unsigned acc = t0 ^ t1 ^ t2 ^ t3 ^ t4 ^ t5 ^ t6 ^ t7 ^ r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7;
    unsigned long long qa = q0 ^ q1 ^ q2 ^ q3 ^ q4 ^ q5 ^ q6 ^ q7;
    double ds = d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7;
    float fs = f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7;
    acc ^= (unsigned)qa ^ (unsigned)(qa >> 32) ^ (unsigned)__double2loint(ds) ^ __float_as_uint(fs);
    if (acc == 0x12345678u) out[0] = acc;      // keep everything live, (almost) never store
    const unsigned long long t_end = __builtin_readcyclecounter();
    if (blockIdx.x == 0 && threadIdx.x == 0) ticks[0] = t_end - t_begin;
}

struct Kind { int id; const char* name; };
static const Kind kinds[] = {
    {0, "v_mad_u64_u32"}, {22, "v_mad_u64_u32(sgpr carry)"}, {1, "v_mul_lo_u32"}, {2, "v_mul_hi_u32"},
    {3, "v_add_u32"}, {4, "v_add_co+v_addc_co"}, {5, "v_add3_u32"}, {6, "v_fma_f64"}, {19, "v_mul_f64"},
    {20, "v_add_f64"}, {21, "v_cvt_f64_u32+v_cvt_u32_f64"}, {7, "v_mad_u32_u24"}, {18, "v_mul_u32_u24"},
    {11, "v_mul_hi_u32_u24"}, {8, "v_lshl_add_u64"}, {9, "v_alignbit_b32"}, {10, "v_lshrrev_b64"},
    {23, "v_bfe_u32"}, {12, "v_fma_f32"}, {17, "v_pk_fma_f32"}, {13, "v_cndmask_b32"}, {14, "v_xor_b32"},
    {15, "mix 1 mad64 : 1 addc"}, {16, "mix 1 mad64 : 3 add"},
    {24, "v_cndmask_b32_e64(sgpr mask)"}, {25, "v_bfi_b32"}, {26, "v_and_or_b32"}, {27, "v_cmp+v_cndmask(vcc)"},
    {28, "v_lshlrev_b32"}, {29, "v_sub_u32"}, {30, "v_and_b32"}, {31, "v_mov_b32"},
    {32, "v_mad_u64_u32(zero addend)"}, {33, "v_mad_u64_u32(one dependent chain)"}, {34, "v_mad_u64_u32(sgpr factor)"},
    {35, "v_mad_i64_i32"}, {43, "v_mad_u64_u32(separate dst)"}, {36, "field column: 12 mad + 2 and + 2 shift64"},
    {37, "v_fma_f32(16 chains, dst != src)"}, {38, "v_add_u32(inline const)"}, {39, "v_pk_fma_f32"},
    {40, "mad64+addc pair (saturated radix)"}, {41, "v_lshl_add_u32"}, {42, "v_or_b32"},
    {44, "v_mad_u64_u32(dependent, ping-pong dst)"}, {45, "v_mad_u64_u32(operands in 4 banks)"},
    {46, "v_mad_u64_u32(operands in 1 bank)"}, {47, "v_mad_u64_u32(src0 == src1)"},
    {48, "1 v_cmp + 16 v_cndmask_b32(vcc)"}, {49, "select by xor/and/xor (per VALU instr)"},
    {50, "s_mov vcc + 16 v_cndmask_b32(vcc)"}, {51, "v_cndmask_b32(vcc) : v_xor alternating"},
    {52, "1 v_cndmask_b32(vcc) : 3 VALU"},
};

static unsigned long long* g_ticks;
template <int KIND> static void launch(int blocks, unsigned* d, hipStream_t s) { k_rate<KIND><<<blocks, 256, 0, s>>>(d, 1, g_ticks); }

static void dispatch(int kind, int blocks, unsigned* d, hipStream_t s)
{
    switch (kind) {
#define C(k) case k: launch<k>(blocks, d, s); break;
        C(0) C(1) C(2) C(3) C(4) C(5) C(6) C(7) C(8) C(9) C(10) C(11) C(12) C(13) C(14) C(15) C(16) C(17) C(18)
        C(19) C(20) C(21) C(22) C(23) C(24) C(25) C(26) C(27) C(28) C(29) C(30) C(31) C(32) C(33) C(34) C(35) C(36)
        C(37) C(38) C(39) C(40) C(41) C(42) C(43) C(44) C(45) C(46) C(47) C(48) C(49) C(50) C(51) C(52)
#undef C
    }
}

int main(int argc, char** argv)
{
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    int cus = prop.multiProcessorCount;
    double ghz = prop.clockRate / 1e6;
    printf("device %s  CUs %d  clock %.3f GHz\n", prop.name, cus, ghz);
    unsigned* d; CHECK(hipMalloc(&d, 64));
    CHECK(hipMalloc(&g_ticks, 8));
    hipStream_t s; CHECK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const char* json = argc > 1 ? argv[1] : nullptr;
    FILE* jf = json ? fopen(json, "w") : nullptr;
    if (jf) fprintf(jf, "{\"device\": \"%s\", \"cus\": %d, \"clock_ghz\": %.3f, \"rates\": {\n", prop.name, cus, ghz);
    bool first = true;
    for (int wps : {1, 2, 4, 8}) {         // waves per SIMD: 256-thread block = 1 wave per SIMD
        int blocks = cus * wps;
        for (const Kind& k : kinds) {
            dispatch(k.id, blocks, d, s);
            CHECK(hipStreamSynchronize(s));
            float best = 1e30f;
            for (int rep = 0; rep < 3; ++rep) {
                CHECK(hipEventRecord(e0, s));
                dispatch(k.id, blocks, d, s);
                CHECK(hipEventRecord(e1, s));
                CHECK(hipEventSynchronize(e1));
                float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
            }
            unsigned long long ticks = 0;
            CHECK(hipMemcpy(&ticks, g_ticks, 8, hipMemcpyDeviceToHost));
            const int per_iter = k.id == 36 ? 16 : UNROLL;                    // instructions per loop iteration
            const double scale = k.id == 39 ? 2.0 : 1.0;                      // packed: two lane-ops per lane
            double insts = (double)blocks * 4 /*waves*/ * ITERS * INNER * per_iter;   // wave-instructions
            double lane_ops_per_s = scale * insts * 64 / (best * 1e-3);
            // wave 0 is the oldest wave of its SIMD and keeps issue priority: `ticks` is how long ITS loop took,
            // i.e. the single-wave issue interval (meaningful at wps = 1, where ticks / time is also the clock)
            double clk_per_inst = (double)ticks / ((double)ITERS * INNER * per_iter);
            double eff_ghz = (double)ticks / (best * 1e-3) / 1e9;
            printf("wps=%d %-42s %8.3f ms  %8.2f Tlane-op/s  %6.2f cycles/instr for the oldest wave  (ticks/time %.2f GHz)\n", wps,
                   k.name, best, lane_ops_per_s / 1e12, clk_per_inst, eff_ghz);
            if (jf) {
                fprintf(jf, "%s  \"%s@wps%d\": %.4e", first ? "" : ",\n", k.name, wps, lane_ops_per_s); first = false;
                fprintf(jf, ",\n  \"cycles:%s@wps%d\": %.3f", k.name, wps, clk_per_inst);
                fprintf(jf, ",\n  \"ghz:%s@wps%d\": %.3f", k.name, wps, eff_ghz);
            }
        }
    }
    if (jf) { fprintf(jf, "\n}}\n"); fclose(jf); }
    return 0;
}
