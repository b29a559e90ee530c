// curve25519_amd/csrc/valu_gfx950.cuh -- the gfx950 instruction sequences the field layer is built from.
//
// Everything above this file (fe25519.cuh, ge25519.cuh, sc25519.cuh, sha512.cuh, x25519.cuh) is plain C++ over
// these primitives, so the CPU unit tests can run that same source against a C model of them
// (tests/host_emul/valu_model.h).  This header is the only one with inline assembly.
//
// Measured issue costs on MI355X, in SIMD cycles per wave-instruction at the kernels' FOUR waves per SIMD (in-kernel s_memtime,
// tools/ubench/mad_peak.hip, profiles/r04_mad_peak.txt; one wave: ~5.1 for everything, two: 4.5):
//   v_mad_u64_u32                      4.26 -- 4.0 of execution and a 0.26 issue bubble -- whether the MADs are independent or
//                                      one dependent column chain (32x32+64 -> 64, carry-out to an SGPR pair)
//   v_lshrrev_b64 / v_mul_lo_u32 / ... 4.26 (every other VOP3 / 64-bit integer instruction)
//   v_add_u32 / v_and_b32 / v_sub_u32  2.13 (VOP2 encodings) when ANOTHER wave's VOP2 shares the issue slot, ~4.2 alone between
//                                      other waves' MADs: hence the low-priority runs below (C25519_VOP2_RUN_*)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#if defined(C25519_CHAIN_DIP) && C25519_CHAIN_DIP               // A/B knob: a priority dip in the middle of a ten-MAD chain
#define C25519_MID_DIP "s_setprio 0\n\ts_setprio 1\n\t"
#else
#define C25519_MID_DIP ""
#endif
#if defined(C25519_MAD_CHAIN_PRIO) && C25519_MAD_CHAIN_PRIO     // A/B knob: only the MAD chains at high wave priority
#define C25519_CHAIN_HI "s_setprio 1\n\t"
#define C25519_CHAIN_LO "\n\ts_setprio 0"
#else
#define C25519_CHAIN_HI ""
#define C25519_CHAIN_LO ""
#endif

namespace c25519 {

typedef uint32_t u32;
typedef uint64_t u64;

#define C25519_DEV __device__ __forceinline__
// nothing is scheduled across this point: keeps the live ranges of two neighbouring field operations apart where the
// scheduler's interleaving would cost registers the kernel does not have (a no-op in the CPU model)
#define C25519_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)

// A run of VOP2 instructions (a field addition / subtraction: ten to twenty 32-bit adds) costs half an issue slot per
// instruction only when ANOTHER wave's VOP2 instruction shares the slot, and a whole slot -- what a MAD costs -- when it
// stands alone among the other waves' MADs (tools/ubench/mad_peak, profiles/r04_mad_peak.txt).  A wave that drops its
// priority for the run is passed over while the others issue MADs and issues its run when a second wave has reached one
// too (or nobody else can issue): the runs pair up.  Measured on the ladder: profiles/r04_ab_prio.txt.
#ifndef C25519_VOP2_RUN_PRIO
#define C25519_VOP2_RUN_PRIO 1        // A/B switch: 0 = no priority changes
#endif
#if C25519_VOP2_RUN_PRIO
#define C25519_VOP2_RUN_BEGIN() __builtin_amdgcn_s_setprio(0)
#define C25519_VOP2_RUN_END() __builtin_amdgcn_s_setprio(1)
#else
#define C25519_VOP2_RUN_BEGIN() do { } while (0)
#define C25519_VOP2_RUN_END() do { } while (0)
#endif

// Where the compiler emits v_mad_u64_u32 for a plain C expression (a 32x32+64 multiply-add outside the asm chains): a
// no-op here; the CPU model of these primitives counts n instructions (tests/host_emul/valu_model.h, tools/executed_macs.py),
// so that the model's count of a ladder step equals the ISA's (739: 5 x 100 + 4 x 55 in chains, + 9 + 10 of these).
#define C25519_COUNT_MAD(n) ((void)0)

// 2x as v_add_u32 x, x: v_add_u32 is full-rate, while v_lshlrev_b32 -- what the compiler picks for x*2 or x+x --
// is in the half-rate class.
C25519_DEV u32 dbl32(u32 x)
{
    u32 r;
    asm("v_add_u32 %0, %1, %1" : "=v"(r) : "v"(x));
    return r;
}

// a / b in single precision, a few ulp off (v_rcp_f32 + v_mul_f32): used only where an estimate is wanted
C25519_DEV float fast_div(float a, float b) { return __fdividef(a, b); }

// Three-input bitwise functions in ONE instruction (v_bitop3_b32, the truth table as an immediate: bit (4a + 2b + c) of it is the
// result for input bits a, b, c): SHA-512's three-way XORs, Ch and Maj -- 24 XORs and 6 ANDs of a round become 12 instructions
// (the compiler does not form them from the two-input source).  Build knob C25519_SHA_BITOP3 = 0: the two-input forms (A/B).
#ifndef C25519_SHA_BITOP3
#define C25519_SHA_BITOP3 1
#endif
#if C25519_SHA_BITOP3
C25519_DEV u32 xor3_32(u32 a, u32 b, u32 c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96); }
C25519_DEV u32 ch_32(u32 e, u32 f, u32 g) { return __builtin_amdgcn_bitop3_b32(e, f, g, 0xca); }       // e ? f : g
C25519_DEV u32 maj_32(u32 a, u32 b, u32 c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0xe8); }
#else
C25519_DEV u32 xor3_32(u32 a, u32 b, u32 c) { return a ^ b ^ c; }
C25519_DEV u32 ch_32(u32 e, u32 f, u32 g) { return (e & f) ^ (~e & g); }
C25519_DEV u32 maj_32(u32 a, u32 b, u32 c) { return (a & b) ^ (a & c) ^ (b & c); }
#endif

// (hi:lo) as one 64-bit value the optimiser cannot take apart again (no instruction: the asm is empty)
C25519_DEV u64 pair64(u32 lo, u32 hi)
{
    u64 r = ((u64)hi << 32) | lo;
    asm("" : "+v"(r));
    return r;
}

// (hi:lo) >> s, low 32 bits   (v_alignbit_b32)
C25519_DEV u32 alignbit32(u32 hi, u32 lo, int s) { return __builtin_amdgcn_alignbit(hi, lo, s); }

// acc + x * y, signed 32 x 32 + 64 -> 64 (v_mad_i64_i32): the matrix applications of safegcd25519.cuh.  The compiler, left to
// itself, turns a signed matrix entry times a masked (known non-negative) limb into v_mad_u64_u32 plus a v_mul_lo_u32 correction.
C25519_DEV int64_t mad_i64_i32(int64_t acc, int32_t x, int32_t y)
{
    u64 carry_out;
    asm("v_mad_i64_i32 %0, %1, %2, %3, %0" : "+v"(acc), "=s"(carry_out) : "v"(x), "v"(y));
    return acc;
}
// acc + x0 * y0 + x1 * y1: one row of a 2 x 2 matrix times a column of limbs, one asm statement (the compiler pads asm boundaries)
C25519_DEV int64_t mad2_i64_i32(int64_t acc, int32_t x0, int32_t y0, int32_t x1, int32_t y1)
{
    u64 carry_out;
    asm("v_mad_i64_i32 %0, %1, %2, %3, %0\n\t"
        "v_mad_i64_i32 %0, %1, %4, %5, %0" : "+v"(acc), "=s"(carry_out) : "v"(x0), "v"(y0), "v"(x1), "v"(y1));
    return acc;
}

// Thirty Bernstein-Yang division steps on low words (safegcd25519.cuh: sg_step_pair is the arithmetic, these are its instruction
// sequences; one asm statement for all thirty, so that no asm boundary is padded and the compiler does not re-derive the step's
// conditions its own, longer way -- 14 instructions a step from the C++ source, 10 here).  At step i, with g0 = the (f, g) pair's Y:
//      c2 = bit i of g0 as a mask (v_bfe_i32)     c1 = zeta < 0 (v_ashrrev_i32)     m = c1 & c2     n = m >> 31
//      per pair:  T = (X ^ c1) & c2 (v_bitop3_b32, truth table 0x28)   Y += T + n (v_add3_u32)   X = (X + (Y & m)) << 1 (v_and_b32, v_add_lshl_u32)
//      zeta = (zeta ^ m) - 1 (v_xad_u32)
#define C25519_SG_COND(I, G0)                    \
    "v_bfe_i32 %[c2], " G0 ", " #I ", 1\n\t"      \
    "v_ashrrev_i32 %[c1], 31, %[zeta]\n\t"        \
    "v_and_b32 %[m], %[c1], %[c2]\n\t"            \
    "v_lshrrev_b32 %[n], 31, %[m]\n\t"
#define C25519_SG_PAIR(X, Y)                                 \
    "v_bitop3_b32 %[t], " X ", %[c1], %[c2] bitop3:0x28\n\t"  \
    "v_add3_u32 " Y ", " Y ", %[t], %[n]\n\t"                 \
    "v_and_b32 %[t], " Y ", %[m]\n\t"                         \
    "v_add_lshl_u32 " X ", " X ", %[t], 1\n\t"
#define C25519_SG_ZETA() "v_xad_u32 %[zeta], %[zeta], %[m], -1\n\t"
#define C25519_SG_STEP1(I) C25519_SG_COND(I, "%[y0]") C25519_SG_PAIR("%[x0]", "%[y0]") C25519_SG_PAIR("%[x1]", "%[y1]") C25519_SG_PAIR("%[x2]", "%[y2]") C25519_SG_ZETA()
#define C25519_SG_STEPQ(I) "v_mov_b32_dpp %[g0], %[y0] quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
    C25519_SG_COND(I, "%[g0]") C25519_SG_PAIR("%[x0]", "%[y0]") C25519_SG_ZETA()
#define C25519_SG_30(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7) S(8) S(9) S(10) S(11) S(12) S(13) S(14) S(15) S(16) S(17) S(18) S(19) \
    S(20) S(21) S(22) S(23) S(24) S(25) S(26) S(27) S(28) S(29)

// one lane: the pairs (f, g) 2^i, (u, q), (v, r)
C25519_DEV void sg_steps30(int32_t& zeta, u32& f, u32& g, u32& u, u32& q, u32& v, u32& r)
{
    u32 c1, c2, m, n, t;
    asm(C25519_SG_30(C25519_SG_STEP1)
        : [zeta] "+v"(zeta), [x0] "+v"(f), [y0] "+v"(g), [x1] "+v"(u), [y1] "+v"(q), [x2] "+v"(v), [y2] "+v"(r),
          [c1] "=&v"(c1), [c2] "=&v"(c2), [m] "=&v"(m), [n] "=&v"(n), [t] "=&v"(t));
}
// a quad of lanes, one pair each; the step's "g odd" comes from lane 0 of the quad.  (The DPP move reads the Y that v_add3_u32
// wrote three instructions earlier -- beyond the two wait states a DPP source needs; the s_nop covers the first step.)
C25519_DEV void sg_steps30_quad(int32_t& zeta, u32& X, u32& Y)
{
    u32 c1, c2, m, n, t, g0;
    asm("s_nop 1\n\t" C25519_SG_30(C25519_SG_STEPQ)
        : [zeta] "+v"(zeta), [x0] "+v"(X), [y0] "+v"(Y),
          [c1] "=&v"(c1), [c2] "=&v"(c2), [m] "=&v"(m), [n] "=&v"(n), [t] "=&v"(t), [g0] "=&v"(g0));
}

// The completion word of a call of ONE element through the host-pointer prototypes (capi_common.hpp: ThreadState::done_word): the
// call's last kernel stores `seq` into pinned host memory BEHIND its results -- by the thread that stored them, or behind a wave's
// own stores: the fence waits for every store of the wave -- and the calling thread, which spins on the word, returns 4.6 us before
// the runtime's event would let it (profiles/r06_launch_latency.txt).  word == nullptr: nobody is waiting that way.  The per-wave
// kernels signal BEFORE they wipe their LDS (coop_ops.cuh): the wipe (1.2-1.6 us) is not the caller's business.
struct DoneWord { u32* word; u32 seq; };
C25519_DEV void signal_done(const DoneWord& d)
{
    if (d.word) {
        __threadfence_system();
        __hip_atomic_store(d.word, d.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// The per-wave kernels' carry of a column sum across the lanes of a 16-lane row (coop25519.cuh: carry(); limb c of a field
// element in lane c): S = l0 + 2^w l1 + 2^51 l2, limb_c = l0_c + l1_(c-1) + l2_(c-2), lanes 9 / 8, 9 wrapping into lanes 0 / 0, 1
// times 19 (m1 / m2: 19 there, 0 elsewhere), then one more single-bit pass.  One asm statement, ordered so that every DPP move
// reads a register written at least two instructions earlier (the hazard the compiler pads with s_nop: three a carry, one here):
// 20 instructions + one s_nop -- a product level of a single call is ~64 instructions, and a call is up to 600 levels in a row.
C25519_DEV u32 row_carry(u64 S, u32 w, u32 mask, u32 mask_next, u32 m1, u32 m2)
{
    u32 limb, l0, l1, l2, a, b;
    asm("v_alignbit_b32 %[l1], %[Shi], %[Slo], %[w]\n\t"           // the low 32 bits of S >> w (w = 25 / 26, per lane)
        "v_lshrrev_b32 %[l2], 19, %[Shi]\n\t"                       // S >> 51
        "v_and_b32 %[l0], %[mask], %[Slo]\n\t"
        "v_and_b32 %[l1], %[maskn], %[l1]\n\t"
        "v_mov_b32_dpp %[a], %[l2] row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_mov_b32_dpp %[b], %[l2] row_ror:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_mul_lo_u32 %[b], %[b], %[m2]\n\t"
        "v_add3_u32 %[limb], %[l0], %[a], %[b]\n\t"
        "v_mov_b32_dpp %[a], %[l1] row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_mov_b32_dpp %[b], %[l1] row_ror:7 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_mul_lo_u32 %[b], %[b], %[m1]\n\t"
        "v_add3_u32 %[limb], %[limb], %[a], %[b]\n\t"
        "v_lshrrev_b32 %[l1], %[w], %[limb]\n\t"                   // e: what the limb holds above its width
        "v_and_b32 %[limb], %[limb], %[mask]\n\t"
        "s_nop 0\n\t"
        "v_mov_b32_dpp %[a], %[l1] row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_mov_b32_dpp %[b], %[l1] row_ror:7 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_mul_lo_u32 %[b], %[b], %[m1]\n\t"
        "v_add3_u32 %[limb], %[limb], %[a], %[b]"
        : [limb] "=&v"(limb), [l0] "=&v"(l0), [l1] "=&v"(l1), [l2] "=&v"(l2), [a] "=&v"(a), [b] "=&v"(b)
        : [Slo] "v"((u32)S), [Shi] "v"((u32)(S >> 32)), [w] "v"(w), [mask] "v"(mask), [maskn] "v"(mask_next), [m1] "v"(m1), [m2] "v"(m2));
    return limb;
}

// acc += sum x[t]*y[t]: one asm statement per column, so the compiler cannot reassociate the chain (it would move
// the carry-in to the end and re-create a separate 64-bit add) and does not pad every MAD with a wait state (it pads
// asm boundaries only).  The SGPR pair receives the (never set) carry-out.
C25519_DEV u64 mad_chain5(u64 acc, const u32 (&x)[5], const u32 (&y)[5])
{
    u64 carry_out;
    asm(
        C25519_CHAIN_HI "v_mad_u64_u32 %0, %1, %2, %7, %0\n\t"
        "v_mad_u64_u32 %0, %1, %3, %8, %0\n\t"
        "v_mad_u64_u32 %0, %1, %4, %9, %0\n\t"
        "v_mad_u64_u32 %0, %1, %5, %10, %0\n\t"
        "v_mad_u64_u32 %0, %1, %6, %11, %0" C25519_CHAIN_LO
        : "+v"(acc), "=s"(carry_out)
        : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]),
          "v"(y[0]), "v"(y[1]), "v"(y[2]), "v"(y[3]), "v"(y[4]));
    return acc;
}

// the same chains started from zero: the first MAD takes the inline constant 0 as its addend instead of an accumulator
// that a v_mov_b64 had to clear (column 0 of every product)
C25519_DEV u64 mad_chain5_from_zero(const u32 (&x)[5], const u32 (&y)[5])
{
    u64 acc, carry_out;
    asm(
        C25519_CHAIN_HI "v_mad_u64_u32 %0, %1, %2, %7, 0\n\t"
        "v_mad_u64_u32 %0, %1, %3, %8, %0\n\t"
        "v_mad_u64_u32 %0, %1, %4, %9, %0\n\t"
        "v_mad_u64_u32 %0, %1, %5, %10, %0\n\t"
        "v_mad_u64_u32 %0, %1, %6, %11, %0" C25519_CHAIN_LO
        : "=&v"(acc), "=s"(carry_out)
        : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]),
          "v"(y[0]), "v"(y[1]), "v"(y[2]), "v"(y[3]), "v"(y[4]));
    return acc;
}

C25519_DEV u64 mad_chain6_from_zero(const u32 (&x)[6], const u32 (&y)[6])
{
    u64 acc, carry_out;
    asm(
        C25519_CHAIN_HI "v_mad_u64_u32 %0, %1, %2, %8, 0\n\t"
        "v_mad_u64_u32 %0, %1, %3, %9, %0\n\t"
        "v_mad_u64_u32 %0, %1, %4, %10, %0\n\t"
        "v_mad_u64_u32 %0, %1, %5, %11, %0\n\t"
        "v_mad_u64_u32 %0, %1, %6, %12, %0\n\t"
        "v_mad_u64_u32 %0, %1, %7, %13, %0" C25519_CHAIN_LO
        : "=&v"(acc), "=s"(carry_out)
        : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]),
          "v"(y[0]), "v"(y[1]), "v"(y[2]), "v"(y[3]), "v"(y[4]), "v"(y[5]));
    return acc;
}

C25519_DEV u64 mad_chain10_from_zero(const u32 (&x)[10], const u32 (&y)[10])
{
    u64 acc, carry_out;
    asm(
        C25519_CHAIN_HI "v_mad_u64_u32 %0, %1, %2, %12, 0\n\t"
        "v_mad_u64_u32 %0, %1, %3, %13, %0\n\t"
        "v_mad_u64_u32 %0, %1, %4, %14, %0\n\t"
        "v_mad_u64_u32 %0, %1, %5, %15, %0\n\t"
        "v_mad_u64_u32 %0, %1, %6, %16, %0\n\t" C25519_MID_DIP
        "v_mad_u64_u32 %0, %1, %7, %17, %0\n\t"
        "v_mad_u64_u32 %0, %1, %8, %18, %0\n\t"
        "v_mad_u64_u32 %0, %1, %9, %19, %0\n\t"
        "v_mad_u64_u32 %0, %1, %10, %20, %0\n\t"
        "v_mad_u64_u32 %0, %1, %11, %21, %0" C25519_CHAIN_LO
        : "=&v"(acc), "=s"(carry_out)
        : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]), "v"(x[8]), "v"(x[9]),
          "v"(y[0]), "v"(y[1]), "v"(y[2]), "v"(y[3]), "v"(y[4]), "v"(y[5]), "v"(y[6]), "v"(y[7]), "v"(y[8]), "v"(y[9]));
    return acc;
}

C25519_DEV u64 mad_chain6(u64 acc, const u32 (&x)[6], const u32 (&y)[6])
{
    u64 carry_out;
    asm(
        C25519_CHAIN_HI "v_mad_u64_u32 %0, %1, %2, %8, %0\n\t"
        "v_mad_u64_u32 %0, %1, %3, %9, %0\n\t"
        "v_mad_u64_u32 %0, %1, %4, %10, %0\n\t"
        "v_mad_u64_u32 %0, %1, %5, %11, %0\n\t"
        "v_mad_u64_u32 %0, %1, %6, %12, %0\n\t"
        "v_mad_u64_u32 %0, %1, %7, %13, %0" C25519_CHAIN_LO
        : "+v"(acc), "=s"(carry_out)
        : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]),
          "v"(y[0]), "v"(y[1]), "v"(y[2]), "v"(y[3]), "v"(y[4]), "v"(y[5]));
    return acc;
}

C25519_DEV u64 mad_chain10(u64 acc, const u32 (&x)[10], const u32 (&y)[10])
{
    u64 carry_out;
    asm(
        C25519_CHAIN_HI "v_mad_u64_u32 %0, %1, %2, %12, %0\n\t"
        "v_mad_u64_u32 %0, %1, %3, %13, %0\n\t"
        "v_mad_u64_u32 %0, %1, %4, %14, %0\n\t"
        "v_mad_u64_u32 %0, %1, %5, %15, %0\n\t"
        "v_mad_u64_u32 %0, %1, %6, %16, %0\n\t" C25519_MID_DIP
        "v_mad_u64_u32 %0, %1, %7, %17, %0\n\t"
        "v_mad_u64_u32 %0, %1, %8, %18, %0\n\t"
        "v_mad_u64_u32 %0, %1, %9, %19, %0\n\t"
        "v_mad_u64_u32 %0, %1, %10, %20, %0\n\t"
        "v_mad_u64_u32 %0, %1, %11, %21, %0" C25519_CHAIN_LO
        : "+v"(acc), "=s"(carry_out)
        : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]), "v"(x[8]), "v"(x[9]),
          "v"(y[0]), "v"(y[1]), "v"(y[2]), "v"(y[3]), "v"(y[4]), "v"(y[5]), "v"(y[6]), "v"(y[7]), "v"(y[8]), "v"(y[9]));
    return acc;
}

}  // namespace c25519
