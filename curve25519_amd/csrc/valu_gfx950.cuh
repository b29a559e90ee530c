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
