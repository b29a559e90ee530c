// curve25519_amd/csrc/quad25519.cuh -- FOUR lanes per element: the shape between "one operation per wave" (coop25519.cuh,
// calls of up to a few thousand elements) and "one element per lane" (x25519.cuh, batches that fill the chip).
//
// A batch of 2^12 .. 2^15 elements, one per lane, is 64 .. 512 waves on a chip of 1024 SIMDs: most of the chip idles while
// every lane walks the whole 0.71 ms chain of a ladder (profiles/r04_batch_sweep.txt: 2^14 elements run at 17 % of the 2^20
// rate).  What the reference's formulas offer to more lanes is the independence of the products INSIDE a step: ecp_Mont
// (source/curve25519_dh.c:57-84) multiplies (x1-z1)(x2+z2), (x2-z2)(x1+z1) and squares (x+z), (x-z) of the point it doubles -- four
// independent products --, then squares the sum and the difference of the first two and multiplies the last two and
// E * (AA + a24 E) -- four more --, and only z3 = x_base * (..)^2 is a third level.  So a QUAD of lanes takes one element:
// every lane holds whole field elements in registers (the fe25519.cuh code, unchanged: same limb bounds, same carry chains),
// each runs ONE fe_mul per level on its own operands, and the operands travel between the four lanes with
// v_mov_b32_dpp quad_perm -- no LDS, no barrier, no wave-wide rendezvous: 16 elements per wave, four times the waves of
// the one-lane kernels.  A step is three product levels (300 MADs in a lane's chain instead of 739) plus ~240 instructions
// of exchange, addition and operand selection: ~1.9 x shorter per element at 4 x the lanes -- right while the chip has
// SIMDs to spare (up to 2^14 elements: one quad-wave per SIMD), wrong once it is full.
//
// Lane q = lane & 3 of a quad carries, between steps,   q0: x of the SUM   q1: z of the sum   q2: x of the DOUBLE   q3: z of the double
// -- the (sum, double) ladder state of x25519.cuh.  Uniform code: all four lanes execute the same instruction stream; what
// differs per lane is which registers a per-lane mask selects as the product's operands.
#pragma once
#include "lanes.cuh"

namespace c25519 {
namespace quad {

// lane q of every quad reads lane P_q of its own quad
template <int P0, int P1, int P2, int P3>
C25519_DEV u32 qperm(u32 x)
{
    return (u32)__builtin_amdgcn_update_dpp(0, (int)x, P0 | (P1 << 2) | (P2 << 4) | (P3 << 6), 0xf, 0xf, true);
}
template <int P0, int P1, int P2, int P3>
C25519_DEV void fe_qperm(fe& r, const fe& a)
{
#pragma unroll
    for (int i = 0; i < 10; i++) r.v[i] = qperm<P0, P1, P2, P3>(a.v[i]);
}

// all-ones on the lane that has the role, zero elsewhere
struct Roles { u32 is0, is1, is2, is3; };
C25519_DEV Roles roles()
{
    const u32 q = threadIdx.x & 3u;
    return Roles{ q == 0 ? 0xffffffffu : 0u, q == 1 ? 0xffffffffu : 0u, q == 2 ? 0xffffffffu : 0u, q == 3 ? 0xffffffffu : 0u };
}

// One ladder step of the element this quad carries.  own: see the role table above; x1: the base point's x (every lane);
// eq: all-ones when this key bit equals the previous one (then the DOUBLE is doubled again, else the sum).
//   level 1   q0: (Sx-Sz)(Dx+Dz)   q1: (Dx-Dz)(Sx+Sz)   q2: P^2   q3: M^2      P, M = x+z, x-z of the point that is doubled
//   level 2   q0: (q0+q1)^2 = x3   q1: (q0-q1)^2        q2: P^2 M^2 = x4   q3: E (P^2 + 121665 E) = z4,  E = P^2 - M^2
//   level 3                        q1: x_base * (..) = z3
// A lane that holds the "wrong" half of a pair computes the negated difference (own - other): it is squared (q1, q3 of
// level 1; q1 of level 2) or meets a second negated factor (q3 of level 2: (-E) * -(P^2 + 121665 E)), so no sign survives.
template <bool BASE9>
C25519_DEV void ladder_step(fe& own, const fe& x1, u32 eq, const Roles& R)
{
    fe other, sum, diff, osum, odiff, X, Y, a, b, p;
    // ---- level 1
    fe_qperm<1, 0, 3, 2>(other, own);                  // the other coordinate of the lane's point
    fe_add(sum, own, other);                           // q0, q1: Sx+Sz          q2, q3: Dx+Dz                      beta 2
    fe_sub(diff, own, other);                          // q0: Sx-Sz  q1: -(Sx-Sz)  q2: Dx-Dz  q3: -(Dx-Dz)          beta 3
    fe_qperm<2, 3, 0, 1>(osum, sum);                   // the other point's x+z
    fe_qperm<2, 2, 0, 0>(odiff, diff);                 // the other point's x-z (from the lane that has it with sign +)
    fe_select(X, R.is1 | (R.is2 & eq), sum, osum);     // q0: Dx+Dz   q1: Sx+Sz   q2: P
    fe_select(Y, R.is0 | (R.is3 & eq), diff, odiff);   // q0: Sx-Sz   q1: Dx-Dz   q3: +-M
    fe_select(a, R.is2, X, Y);
    fe_select(b, R.is3, Y, X);
    fe_mul(p, a, b);
    // ---- level 2
    fe_qperm<1, 0, 3, 2>(other, p);
    fe_add(sum, p, other);                             // q0, q1: DA+CB                                              beta 2
    fe_sub(diff, p, other);                            // q0: DA-CB  q1: -(DA-CB)  q2: E = AA-BB  q3: -E            beta 3
    fe_neg(X, other);                                  // q3: -AA                                                   beta 2
    fe_mul121665_add(Y, X, diff);                      // q3: -(AA + 121665 E), reduced
    fe_select(X, R.is0, sum, diff);
    fe_select(a, R.is2, p, X);                         // q0: sum    q1: diff   q2: AA   q3: -E
    fe_select(b, R.is3, Y, X);
    fe_select(b, R.is2, other, b);                     // q0: sum    q1: diff   q2: BB   q3: -(AA + 121665 E)
    fe_mul(p, a, b);
    // ---- level 3: z3 = x_base * (DA-CB)^2 on q1; the other lanes keep their level-2 product
    if (BASE9) fe_mul_small(a, p, 9);
    else fe_mul(a, p, x1);
    fe_select(own, R.is1, a, p);
}

// curve25519_dh_CreateSharedKey / _CalculatePublicKey (BASE9) for element e, by the quad this lane belongs to: the
// reference's bytes (ecp_PointMultiply, curve25519_dh.c:94-157: clamped key written back, all 256 bits of the peer key
// used, a zero Z gives zero bytes).  Every lane of the quad loads the element's records; lane 0 of the quad stores.
// The first doubling (Q = 2P, :125), the three doublings for the clamped-away low bits and the inversion have no
// four-way structure worth an exchange: every lane runs them on the same values (the inversion is 12 % of the element's
// chain -- Montgomery's trick would share it between the 16 elements of the wave but not shorten the chain).
template <bool BASE9>
C25519_DEV void x25519_element(void* out, const void* pk, void* sk, size_t e)
{
    const Roles R = roles();
    u32 u[8] = { 9, 0, 0, 0, 0, 0, 0, 0 }, k[8];
    if (!BASE9) load32(u, pk, e);
    load32(k, sk, e);
    clamp_words(k);
    if (R.is0) store32(sk, e, k);                      // the reference clamps in the caller's buffer
    fe X1, own;
    fe_from_words(X1, u);
    {
        fe DX = X1, DZ, one, t0, t1;
        fe_set_u32(one, 1);
        DZ = one;
        mont_double(DX, DZ);
        fe_select(t0, R.is0, X1, one);
        fe_select(t1, R.is2, DX, DZ);
        fe_select(own, R.is0 | R.is1, t0, t1);
    }
    // the bit scan of x25519_ladder_xz (x25519.cuh): bit 254 is the leading one, bits 2..0 are zero after clamping
    u32 prev = 1;
    u32 kq[8];
#pragma unroll
    for (int t = 0; t < 8; t++) kq[t] = k[t];
#pragma unroll 1
    for (int w = 7; w >= 0; w--) {
        u32 kw = kq[7];
#pragma unroll
        for (int t = 7; t > 0; t--) kq[t] = kq[t - 1];
        const int top = (w == 7) ? 29 : 31;
        const int bottom = (w == 0) ? 3 : 0;
        kw <<= (31 - top);
#pragma unroll 1
        for (int bit_no = top; bit_no >= bottom; bit_no--) {
            const u32 bit = kw >> 31;
            kw <<= 1;
            ladder_step<BASE9>(own, X1, (u32)0 - (u32)(bit == prev), R);
            prev = bit;
        }
    }
    fe PX, PZ;
    {
        fe SX, SZ, DX, DZ;
        fe_qperm<0, 0, 0, 0>(SX, own);
        fe_qperm<1, 1, 1, 1>(SZ, own);
        fe_qperm<2, 2, 2, 2>(DX, own);
        fe_qperm<3, 3, 3, 3>(DZ, own);
        const u32 m = (u32)0 - prev;                   // P = S if the last bit was 1, else D (curve25519_dh.c:148-150)
        fe_select(PX, m, SX, DX);
        fe_select(PZ, m, SZ, DZ);
    }
#pragma unroll 1
    for (int i = 0; i < 3; i++) mont_double(PX, PZ);
    fe zi;
    fe_invert(zi, PZ);                                 // z = 0 gives 0, like the reference's z^(p-2)
    fe_mul(PX, PX, zi);
    u32 w[8];
    fe_to_words(w, PX);
    if (R.is0) store32(out, e, w);                     // written last: `out` may alias `pk`
}

constexpr int ELEMS_PER_WAVE = 16;

}  // namespace quad
}  // namespace c25519
