// curve25519_amd/csrc/x25519.cuh -- variable-base Montgomery ladder, one keypair per lane.
//
// Device replacement for ecp_PointMultiply (source/curve25519_dh.c:94-157) as reached through
// curve25519_dh_CreateSharedKey (:201-208) and curve25519_dh_CalculatePublicKey (:191-198):
//   * the secret key is clamped (ecp_TrimSecretKey, curve25519_utils.c:28) and the clamped bytes are
//     written back, because the reference clamps in the caller's buffer;
//   * all 256 bits of the peer public key are used, reduced mod p (no masking of bit 255, no
//     low-order rejection: those inputs produce 32 zero bytes exactly like the reference);
//   * after clamping the top set bit is always bit 254, so the ladder is 254 fixed steps;
//   * the reference's random projective Z (:123) is output-neutral; Z = 1 here.
//
// The reference selects (P,Q) by pointer swap (ECP_MONT, :89).  Here the differential addition is
// symmetric in its two inputs, so the state is kept as (S = sum, D = double) and only the INPUT OF
// THE DOUBLING is selected per bit (the compiler emits v_cndmask_b32) -- 20 selects per step instead of a 40-limb swap,
// no secret-dependent branch or address.
#pragma once
#include "fe25519.cuh"

namespace c25519 {

// one ladder step.  prev_eq = all-ones when this bit equals the previous one.
//   S' = S + D (difference = base),  D' = 2 * (prev_eq ? D : S)
// (X : Z) <- 2 (X : Z)   (ecp_MontDouble, curve25519_dh.c:40-54)
C25519_DEV void mont_double(fe& X, fe& Z)
{
    fe A, B;
    fe_add(A, X, Z);                   // beta 2
    fe_sub(B, X, Z);                   // beta 3
    fe_sqr_runs(A, A);
    fe_sqr_runs(B, B);
    fe_mul_runs(X, A, B);
    fe_sub(B, A, B);                   // beta 3
    fe_mul121665_add(A, A, B);
    fe_mul_runs(Z, B, A);
}

#ifndef C25519_LADDER_PRIO_SITES
#define C25519_LADDER_PRIO_SITES 0
#endif

// no-op section marker of the product build; the opt-in cycle-probe build (engine.hip, -DC25519_CYCLE_PROBE=2) passes
// one that reads s_memtime, so that the sections of a step can be timed in place (tools/cycle_probe.py)
struct NoSectionMark { C25519_DEV void operator()(int) const {} };

// BASE9: the difference point is the curve's base point u = 9 (curve25519_dh_CalculatePublicKey), so the
// one multiplication by it is a 10-MAD small-constant multiply instead of a full product.
template <bool BASE9 = false, typename Mark = NoSectionMark>
C25519_DEV void ladder_step(fe& SX, fe& SZ, fe& DX, fe& DZ, const fe& base, u32 prev_eq, Mark mark = Mark())
{
    fe A, B, C, Dp, P, M;
#if C25519_LADDER_PRIO_SITES >= 1
    C25519_VOP2_RUN_BEGIN();
#endif
    fe_sub(A, SX, SZ);                 // beta 3
    fe_add(B, SX, SZ);                 // beta 2
    fe_sub(C, DX, DZ);                 // beta 3
    fe_add(Dp, DX, DZ);                // beta 2
    fe_select(P, prev_eq, Dp, B);      // doubling input, x+z
    fe_select(M, prev_eq, C, A);       // doubling input, x-z
#if C25519_LADDER_PRIO_SITES >= 1
    C25519_VOP2_RUN_END();
#endif
    mark(0);
    fe_mul_runs(A, A, Dp);                  // (x1-z1)(x2+z2)
    mark(1);
    fe_mul_runs(B, C, B);                   // (x2-z2)(x1+z1)
    mark(2);
#if C25519_LADDER_PRIO_SITES >= 2
    C25519_VOP2_RUN_BEGIN();
#endif
    fe_add(C, A, B);                   // beta 2
    fe_sub(B, A, B);                   // beta 3
#if C25519_LADDER_PRIO_SITES >= 2
    C25519_VOP2_RUN_END();
#endif
    mark(3);
    fe_sqr_runs(SX, C);                     // x3
    fe_sqr_runs(A, B);
    mark(4);
    if (BASE9) fe_mul_small(SZ, A, 9);  // z3 = (..)^2 * 9
    else fe_mul_runs(SZ, A, base);          // z3 = (..)^2 * xb
    mark(5);
    fe_sqr_runs(A, P);                      // (x+z)^2
    fe_sqr_runs(B, M);                      // (x-z)^2
    mark(6);
    fe_mul_runs(DX, A, B);                  // x4
    mark(7);
#if C25519_LADDER_PRIO_SITES >= 3
    C25519_VOP2_RUN_BEGIN();
#endif
    fe_sub(B, A, B);                   // beta 3
#if C25519_LADDER_PRIO_SITES >= 3
    C25519_VOP2_RUN_END();
#endif
    fe_mul121665_add(A, A, B);         // (x+z)^2 + 121665*B
    mark(8);
    fe_mul_runs(DZ, B, A);                  // z4
    mark(9);
}

// (PX : PZ) = clamp(k) * (u : 1), x-only, projective.  k are the CLAMPED scalar words.  The affine result
// PX/PZ is produced by the shared batched-inversion kernel (engine.hip), which amortises ecp_Inverse
// (curve25519_dh.c:148) over several elements.
template <bool BASE9 = false, typename Mark = NoSectionMark>
C25519_DEV void x25519_ladder_xz(fe& PX, fe& PZ, const u32 (&u)[8], const u32 (&k)[8], Mark mark = Mark())
{
    fe X1, SX, SZ, DX, DZ;
    fe_from_words(X1, u);

    // P = (X1 : 1), Q = 2P  (curve25519_dh.c:123-125 with zr = 1); bit 254 is the leading one
    SX = X1;
    fe_set_u32(SZ, 1);
    DX = SX;
    DZ = SZ;
    mont_double(DX, DZ);

    // state invariant: previous bit b_prev = 1  <=>  (P,Q) = (S,D); else (P,Q) = (D,S)
    u32 prev = 1;
    u32 kq[8];                                           // the scalar's words as a queue: the next one is always kq[7] (an index
#pragma unroll                                           // by the loop counter, or a chain of selects on it, ends up as a
    for (int t = 0; t < 8; t++) kq[t] = k[t];            // scratch array -- the key would go through memory)
#pragma unroll 1
    for (int w = 7; w >= 0; w--) {
        u32 kw = kq[7];
#pragma unroll
        for (int t = 7; t > 0; t--) kq[t] = kq[t - 1];
        const int top = (w == 7) ? 29 : 31;              // bit 254 consumed above, bit 255 is zero
        const int bottom = (w == 0) ? 3 : 0;             // bits 2..0 are zero after clamping: handled below
        kw <<= (31 - top);
#pragma unroll 1
        for (int b = top; b >= bottom; b--) {
            const u32 bit = kw >> 31;
            kw <<= 1;
            const u32 eq = (u32)0 - (u32)(bit == prev);
            mark(-1);                                    // step entry
            ladder_step<BASE9>(SX, SZ, DX, DZ, X1, eq, mark);
            prev = bit;
        }
    }
    // P = PP[1] is what the reference converts (:148-150): P = S if the last bit was 1, else D
    const u32 m = (u32)0 - prev;
    fe_select(PX, m, SX, DX);
    fe_select(PZ, m, SZ, DZ);
    // the three clamped-away low bits: a zero bit maps (P, Q) to (2P, P+Q) and only P is ever read again, so
    // the reference's last three ecp_Mont calls reduce to three doublings of P
#pragma unroll 1
    for (int i = 0; i < 3; i++) mont_double(PX, PZ);
}

}  // namespace c25519
