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
#include "verify_fast.cuh"

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

// additions, subtractions and selects without the wave-priority dips of fe25519.cuh's (a quad-wave has its SIMD to itself:
// there is no other wave's VOP2 run to pair up with)
C25519_DEV void q_add(fe& r, const fe& a, const fe& b)
{
#pragma unroll
    for (int i = 0; i < 10; i++) r.v[i] = a.v[i] + b.v[i];
}
C25519_DEV void q_sub(fe& r, const fe& a, const fe& b)            // a - b + 2p, b reduced
{
#pragma unroll
    for (int i = 0; i < 10; i++) r.v[i] = a.v[i] + fe_2p(i) - b.v[i];
}
C25519_DEV void q_neg(fe& r, const fe& a)
{
#pragma unroll
    for (int i = 0; i < 10; i++) r.v[i] = fe_2p(i) - a.v[i];
}
C25519_DEV void q_sel(fe& r, u32 mask, const fe& a, const fe& b)  // mask ? a : b
{
#pragma unroll
    for (int i = 0; i < 10; i++) r.v[i] = (a.v[i] & mask) | (b.v[i] & ~mask);
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
    fe other, sum, diff, give, keep, got, a, b, p;
    // ---- level 1
    fe_qperm<1, 0, 3, 2>(other, own);                  // the other coordinate of the lane's point
    q_add(sum, own, other);                            // q0, q1: Sx+Sz          q2, q3: Dx+Dz                      beta 2
    q_sub(diff, own, other);                           // q0: Sx-Sz  q1: -(Sx-Sz)  q2: Dx-Dz  q3: -(Dx-Dz)          beta 3
    q_sel(give, R.is1 | R.is2, diff, sum);             // what the other pair wants of this lane: q0: Sx+Sz  q1: -(Sx-Sz)  q2: Dx-Dz  q3: Dx+Dz
    q_sel(keep, R.is1 | R.is2, sum, diff);             // ... and the lane's own factor:        q0: Sx-Sz  q1: Sx+Sz     q2: Dx+Dz  q3: -(Dx-Dz)
    fe_qperm<3, 2, 0, 1>(got, give);                   // q0: Dx+Dz   q1: Dx-Dz   q2: Sx+Sz   q3: -(Sx-Sz)
    const u32 twice = (R.is2 | R.is3) & eq;            // the doubling lanes square their own point's P, M when it is doubled again
    q_sel(a, R.is0 | twice, keep, got);                // q0: Sx-Sz   q1: Dx-Dz   q2: P   q3: +-M
    q_sel(b, R.is1 | twice, keep, got);                // q0: Dx+Dz   q1: Sx+Sz   q2: P   q3: +-M
    fe_mul(p, a, b);
    // ---- level 2
    fe_qperm<1, 0, 3, 2>(other, p);
    q_add(sum, p, other);                              // q0, q1: DA+CB                                              beta 2
    q_sub(diff, p, other);                             // q0: DA-CB  q1: -(DA-CB)  q2: E = AA-BB  q3: -E            beta 3
    q_neg(keep, other);                                // q3: -AA                                                   beta 2
    fe_mul121665_add(got, keep, diff);                 // q3: -(AA + 121665 E), reduced
    q_sel(give, R.is0, sum, diff);
    q_sel(a, R.is2, p, give);                          // q0: sum    q1: diff   q2: AA   q3: -E
    q_sel(b, R.is3, got, give);
    q_sel(b, R.is2, other, b);                         // q0: sum    q1: diff   q2: BB   q3: -(AA + 121665 E)
    fe_mul(p, a, b);
    // ---- level 3: z3 = x_base * (DA-CB)^2 on q1; the other lanes keep their level-2 product
    if (BASE9) fe_mul_small(a, p, 9);
    else fe_mul(a, p, x1);
    q_sel(own, R.is1, a, p);
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
    fe_invert_quad(zi, PZ);                            // z = 0 gives 0, like the reference's z^(p-2)
    fe_mul(PX, PX, zi);
    u32 w[8];
    fe_to_words(w, PX);
    if (R.is0) store32(out, e, w);                     // written last: `out` may alias `pk`
}

constexpr int ELEMS_PER_WAVE = 16;

// =====================================================================================================================
// Edwards points on a quad.  Lane roles between operations:   q0: X   q1: Y   q2: T   q3: Z   (extended coordinates, all reduced)
// The reference's formulas (edp_AddPoint source/ed25519_verify.c:142-161, edp_AddAffinePoint / edp_DoublePoint
// source/ed25519_sign.c:97-143; ge25519.cuh has them one point per lane) are two levels of four independent products each:
//   addition   level 1: B = (Y+X) ypx   A = (Y-X) ymx   C = T t2d   D = Z z2        -- one field of the table row per lane
//              level 2: X3 = e f   Y3 = h g   T3 = e h   Z3 = f g     (e = B-A, h = B+A, f = D-C, g = D+C)
//   doubling   level 1: A = X^2   B = Y^2   (X+Y)^2   C = Z^2                       -- four squarings: fe_sqr, 55 MADs
//              level 2: X3 = E Fn   Y3 = G Hn   T3 = E Hn   Z3 = G Fn  (Hn = A+B, G = B-A, E = (X+Y)^2 - Hn, Fn = 2C - G)
// so a quad runs an addition in two product levels instead of eight products in a row, a doubling in a squaring and a product
// instead of four and four.  Sums and differences are formed inside a pair of lanes (X with Y, T with Z: one quad_perm swap),
// the cross terms fetched from the other pair; where a lane holds the negated difference it is not used (each product has a
// lane that holds both its factors with the right sign, or fetches them).
// =====================================================================================================================

// own <- own + P, `mult` = the lane's field of P's precomputed row: q0: Y+X (of -P: Y-X)   q1: Y-X (Y+X)   q2: 2dT (-2dT)
// q3: 2Z (an affine row: the constant 2).  mult: beta <= 2.
C25519_DEV void ge_add_fields(fe& own, const fe& mult, const Roles& R)
{
    fe other, sum, diff, a, b, p, f1, f2;
    fe_qperm<1, 0, 3, 2>(other, own);
    q_add(sum, own, other);                            // q0, q1: Y+X                                                beta 2
    q_sub(diff, own, other);                           // q1: Y-X                                                    beta 3
    q_sel(a, R.is0, sum, own);
    q_sel(a, R.is1, diff, a);                          // q0: Y+X   q1: Y-X   q2: T   q3: Z
    fe_mul(p, a, mult);                                // q0: B     q1: A     q2: C   q3: D
    fe_qperm<1, 0, 3, 2>(other, p);
    q_add(sum, p, other);                              // q0, q1: h = B+A     q2, q3: g = D+C                        beta 2
    q_sub(diff, p, other);                             // q0: e = B-A   q1: -e   q2: -f   q3: f = D-C                beta 3
    fe_qperm<3, 1, 0, 3>(f1, diff);                    // q0: f     q2: e
    fe_qperm<0, 2, 0, 3>(f2, sum);                     // q1: g     q2: h
    q_sel(a, R.is2, f1, diff);
    q_sel(a, R.is1, sum, a);                           // q0: e   q1: h   q2: e   q3: f
    q_sel(b, R.is3, sum, f2);
    q_sel(b, R.is0, f1, b);                            // q0: f   q1: g   q2: h   q3: g
    fe_mul(own, a, b);                                 // q0: X3 = e f   q1: Y3 = h g   q2: T3 = e h   q3: Z3 = f g
}

// own <- 2 own   (T is not read)
C25519_DEV void ge_double(fe& own, const Roles& R)
{
    fe other, sum, diff, a, b, p, F, u, U2, U3;
    fe_qperm<1, 0, 3, 2>(other, own);
    q_add(sum, own, other);                            // q0: X+Y                                                    beta 2
    fe_qperm<0, 1, 0, 3>(F, sum);
    q_sel(a, R.is2, F, own);                           // q0: X   q1: Y   q2: X+Y   q3: Z
    fe_sqr(p, a);                                      // q0: A   q1: B   q2: (X+Y)^2   q3: C
    fe_qperm<1, 0, 3, 2>(other, p);
    q_add(sum, p, other);                              // q0, q1: Hn = A+B                                           beta 2
    q_sub(diff, p, other);                             // q0: -G   q1: G = B-A                                       beta 3
    q_sel(a, R.is1, diff, sum);
    fe_qperm<0, 1, 0, 1>(F, a);                        // q2: Hn   q3: G
    // q2: E = (X+Y)^2 - Hn   q3: Fn = 2C - G   (bias 4p: the subtrahends have beta 2 and 3), carried back to reduced -- E Fn
    // needs one of them reduced, and both are second factors below
#pragma unroll
    for (int i = 0; i < 10; i++) u.v[i] = p.v[i] + (p.v[i] & R.is3) + 2u * fe_2p(i) - F.v[i];
    fe_carry32(u, u);
    fe_qperm<2, 1, 2, 3>(U2, u);                       // q0: E
    fe_qperm<3, 1, 2, 3>(U3, u);                       // q0: Fn
    q_sel(a, R.is1, diff, F);
    q_sel(a, R.is0, U2, a);                            // q0: E    q1: G    q2: Hn   q3: G
    q_sel(b, R.is1, sum, u);
    q_sel(b, R.is0, U3, b);                            // q0: Fn   q1: Hn   q2: E    q3: Fn
    fe_mul(own, a, b);                                 // q0: X3 = E Fn   q1: Y3 = G Hn   q2: T3 = Hn E   q3: Z3 = G Fn
}

// the neutral element (0 : 1 : 0 : 1)
C25519_DEV void ge_neutral(fe& own, const Roles& R)
{
    fe_set_u32(own, 0);
    own.v[0] = (R.is1 | R.is3) & 1u;
}

// ---- the lane's field of a packed table row (verify_fast.cuh: (Y+X | Y-X | 2dT | 2Z), 8 words each), fetched early, unpacked late
struct field_raw { uint4 lo, hi; u32 neg; };
C25519_DEV void row_field_fetch(field_raw& r, const u32* row, u32 neg)
{
    const u32 q = threadIdx.x & 3u;
    const u32 fld = q < 2u ? (q ^ (neg & 1u)) : q;      // -P: Y+X and Y-X trade places
    const uint4* p = reinterpret_cast<const uint4*>(row + 8 * fld);
    r.lo = p[0];
    r.hi = p[1];
    r.neg = neg;
}
C25519_DEV void row_field_unpack(fe& mult, const field_raw& r, const Roles& R)
{
    const u32 w[8] = { r.lo.x, r.lo.y, r.lo.z, r.lo.w, r.hi.x, r.hi.y, r.hi.z, r.hi.w };
    fe f, n;
    fe_from_words(f, w);
    q_neg(n, f);                                        // -P: 2p - 2dT, beta 2
    q_sel(mult, R.is2 & r.neg, n, f);
}

// the lane's field of column c of the walk's signed comb (limb-major LDS table [30][SC_ROWS], ge_add_pa_comb's recoding): an
// affine row, so q3's factor is the constant 2
C25519_DEV void comb_field(fe& mult, const u32* tbl, u32 c, const Roles& R)
{
    const u32 neg = ((c >> (SC_TEETH - 1)) & 1u) - 1u;   // all-ones: negative column
    const u32 r = (c ^ neg) & (u32)(SC_ROWS - 1);
    const u32 q = threadIdx.x & 3u;
    const u32 fld = q < 2u ? (q ^ (neg & 1u)) : 2u;
    const u32* p = tbl + 10 * fld * SC_ROWS + r;
    fe f, n, two;
#pragma unroll
    for (int i = 0; i < 10; i++) f.v[i] = p[i * SC_ROWS];
    q_neg(n, f);
    q_sel(f, R.is2 & neg, n, f);
    fe_set_u32(two, 2);
    q_sel(mult, R.is3, two, f);
}

// ge_walk_is_neutral (verify_fast.cuh) by a quad: W = sigma*B + tau*Q + rho*Rn from the element's two window tables, its biased
// scalars and the LDS comb; all-ones iff W is the neutral element.  Same digits, same rows, same order of operations; the walk
// starts from the neutral element (one more addition than the one-lane walk's "first row as the starting point").
// q_flip (all-ones or zero): the table at tq holds the point as decoded and the walk wants its negative (tau < 0: the rows'
// signs flip, nothing else -- the points were decoded and tabulated beside the scalar work, before tau's sign was known).
C25519_DEV u32 walk_is_neutral(const WalkScalars& sc, const u32* tq, const u32* tr, const u32* lds_tbl, int top, const Roles& R,
                               u32 q_flip = 0)
{
    fe own, mult;
    ge_neutral(own, R);
    {
        u32 neg;
        field_raw rq, rr;
        const u32 mq = signed16_of(neg, sc.tau_word(top >> 3), top & 7);
        row_field_fetch(rq, tq + mq * ROW_WORDS, neg ^ q_flip);
        const u32 mr = signed16_of(neg, sc.rho_word(top >> 3), top & 7);
        row_field_fetch(rr, tr + mr * ROW_WORDS, neg);
        row_field_unpack(mult, rq, R);
        ge_add_fields(own, mult, R);
        row_field_unpack(mult, rr, R);
        ge_add_fields(own, mult, R);
    }
#pragma unroll 1
    for (int i = top - 1; i >= 0; i--) {
        const u32 tw = sc.tau_word(i >> 3), rw = sc.rho_word(i >> 3);
        u32 neg;
        field_raw rq, rr;                               // the round's two rows: in flight under the doublings
        const u32 mq = signed16_of(neg, tw, i & 7);
        row_field_fetch(rq, tq + mq * ROW_WORDS, neg ^ q_flip);
        const u32 mr = signed16_of(neg, rw, i & 7);
        row_field_fetch(rr, tr + mr * ROW_WORDS, neg);
        if (i >= SC_ROUNDS) {
#pragma unroll 1
            for (int j = 0; j < 4; j++) ge_double(own, R);
        } else {
            u64 cols = (u64)sc.sigma_word(2 * i) | ((u64)sc.sigma_word(2 * i + 1) << 32);
#pragma unroll 1
            for (int j = 0; j < 4; j++) {
                ge_double(own, R);
                if (4 * i + 3 - j < SC_COLS) {          // wave-uniform: the first round may carry fewer than four
                    comb_field(mult, lds_tbl, (u32)cols & 0xffffu, R);
                    ge_add_fields(own, mult, R);
                }
                cols >>= 16;
            }
        }
        row_field_unpack(mult, rq, R);
        ge_add_fields(own, mult, R);
        row_field_unpack(mult, rr, R);
        ge_add_fields(own, mult, R);
    }
    // neutral element: X == 0 and Y == Z (Z != 0 for on-curve inputs under the complete law)
    fe X, Y, Z, d;
    fe_qperm<0, 0, 0, 0>(X, own);
    fe_qperm<1, 1, 1, 1>(Y, own);
    fe_qperm<3, 3, 3, 3>(Z, own);
    u32 xw[8], dw[8], acc = 0;
    q_sub(d, Y, Z);
    fe_to_words(xw, X);
    fe_to_words(dw, d);
#pragma unroll
    for (int i = 0; i < 8; i++) acc |= xw[i] | dw[i];
    return acc == 0 ? 0xffffffffu : 0u;
}


// =====================================================================================================================
// Fixed-base multiples k * B over the WIDE comb (ge25519.cuh: ge_base_mult_wide -- 13 signed teeth, four tables of 4096 packed
// 128-byte rows read through L2) by a quad: what ed25519_CreateKeyPair / ed25519_SignMessage / curve25519_dh_CalculatePublicKey_fast
// calls of 2^11 .. 2^14 elements run (engine.hip: k_ed25519_*_quad).  One lane per element walks 19 additions of seven products and
// four doublings of eight, one after the other (~22 000 instructions), then waits for two more launches (the shared inversion,
// the last hash); a quad walks 20 additions of TWO product levels and the doublings as a level of squarings and one of products
// (~8 500), and carries on in the same launch: inversion, encoding, h and S = h a + r by all four lanes on the same values
// (hashing has no four-way structure; the chip has the SIMDs to spare at these sizes).  A row of the comb is the precomputed
// form of an affine point: (Y+X | Y-X | 2dT | 2) -- the fourth field, the constant 2 = 2Z, is the row's padding, so lane q's
// factor is word group q of the line whatever the point (row_field_fetch).  The walk starts from the neutral element (one
// addition more than ge_base_mult_wide's "first row as the starting point": the formulas are complete).
// =====================================================================================================================

// the lane's field of the row column c (13 bits: tooth 12 is the sign) selects in `tbl`
C25519_DEV void wide_field_fetch(field_raw& r, const u32* __restrict__ tbl, u32 c)
{
    const u32 neg = ((c >> (WB_TEETH - 1)) & 1u) - 1u;          // all-ones: negative column
    const u32 row = (c ^ neg) & (u32)(WB_ROWS - 1);
    row_field_fetch(r, tbl + (size_t)row * WB_ROW_WORDS, neg);
}

// own <- k * B.  cols: this lane's parked columns of k (wb_columns; every lane of the quad parks the same twenty), `stride`
// elements apart, in the order the walk consumes them: s = m * WB_NT + t selects a row of table t; a doubling in front of every
// m > 0.  The next row is fetched under the current addition (~1 600 cycles against ~700 of L2 latency).
C25519_DEV void base_mult_wide(fe& own, const u32* __restrict__ g_wide, const unsigned short* cols, int stride, const Roles& R)
{
    static_assert(ROW_WORDS == WB_ROW_WORDS, "one row shape for the window tables and the wide comb");
    fe mult;
    field_raw cur, nxt;
    ge_neutral(own, R);
    wide_field_fetch(cur, g_wide, cols[0]);
    nxt = cur;
#pragma unroll 1
    for (int s = 0; s < WB_COLS; s++) {
        const int t = s & (WB_NT - 1);
        if (s + 1 < WB_COLS)
            wide_field_fetch(nxt, g_wide + (size_t)((s + 1) & (WB_NT - 1)) * WB_ROWS * WB_ROW_WORDS, cols[(s + 1) * stride]);
        if (t == 0 && s) ge_double(own, R);
        row_field_unpack(mult, cur, R);
        ge_add_fields(own, mult, R);
        cur = nxt;
    }
}

// own <- s * B + h * P over TWO wide combs walked together (ge25519.cuh: ge_double_base_mult_wide; the two-phase verification's
// key comb, engine_verify.hip): 40 additions from the neutral element and the same 4 doublings, the rows of the base point's and
// of P's tables in turn, the next row fetched under the current addition.  colsB / colsP: this lane's parked columns of s and h.
C25519_DEV void double_base_mult_wide(fe& own, const u32* __restrict__ wideB, const unsigned short* colsB, const u32* __restrict__ wideP,
                                      const unsigned short* colsP, int stride, const Roles& R)
{
    fe mult;
    field_raw cur, nxt;
    ge_neutral(own, R);
    wide_field_fetch(cur, wideB, colsB[0]);
    nxt = cur;
#pragma unroll 1
    for (int k = 0; k < 2 * WB_COLS; k++) {
        const int s = k >> 1, t = s & (WB_NT - 1);
        if (k + 1 < 2 * WB_COLS) {
            const int s1 = (k + 1) >> 1, t1 = s1 & (WB_NT - 1);
            const u32* tb = ((k + 1) & 1) ? wideP : wideB;
            const unsigned short* cc = ((k + 1) & 1) ? colsP : colsB;
            wide_field_fetch(nxt, tb + (size_t)t1 * WB_ROWS * WB_ROW_WORDS, cc[s1 * stride]);
        }
        if (!(k & 1) && t == 0 && s) ge_double(own, R);
        row_field_unpack(mult, cur, R);
        ge_add_fields(own, mult, R);
        cur = nxt;
    }
}

// enc(P) (ed25519_PackPoint, curve25519_utils.c:77-98: y with the parity of x in bit 255) of own = (X, Y, T, Z), in every lane.
// One inversion (ed25519_sign.c:265) by all four lanes; then x on q0 and y on q1 are one product.
C25519_DEV void encode_point(u32 (&enc)[8], const fe& own)
{
    fe Z, zi, a;
    fe_qperm<3, 3, 3, 3>(Z, own);
    fe_invert_quad(zi, Z);
    fe_mul(a, own, zi);                                    // q0: x   q1: y
    u32 w[8];
    fe_to_words(w, a);
    const u32 x_odd = qperm<0, 0, 0, 0>(w[0]) << 31;
#pragma unroll
    for (int i = 0; i < 8; i++) enc[i] = qperm<1, 1, 1, 1>(w[i]);
    enc[7] = (enc[7] & 0x7fffffffu) | x_odd;
}

// ed25519_CreateKeyPair (ed25519_sign.c:344-367) for element e: priv = sk || enc(A), pub = enc(A), A = clamp(H(sk)[0..31]) * B
C25519_DEV void keypair_element(void* pub, void* priv, const void* sk, size_t e, const u32* __restrict__ g_wide,
                                unsigned short* cols, int stride)
{
    const Roles R = roles();
    u32 seed[8], a[8], enc[8];
    u64 b_words[4];
    load32(seed, sk, e);
    ed_expand_seed(a, b_words, seed);
    wb_columns(cols, stride, a);
    fe own;
    base_mult_wide(own, g_wide, cols, stride, R);
    encode_point(enc, own);
    if (R.is0) {
        store32(priv, 2 * e, seed);
        store32(priv, 2 * e + 1, enc);
        store32(pub, e, enc);
    }
}

// ed25519_SignMessage (ed25519_sign.c:372-419) for element e: r = H(H(sk)[32..63] || m) mod L, R = r * B, S = H(enc(R) || pk || m) a + r
C25519_DEV void sign_element(void* sig, const void* priv, const uint8_t* msg, size_t len, size_t e, const u32* __restrict__ g_wide,
                             unsigned short* cols, int stride)
{
    const Roles R = roles();
    u32 seed[8], a[8], r[8], encR[8], pkw[8], s[8];
    load32(seed, priv, 2 * e);
    ed_sign_nonce(a, r, seed, msg, len);
    wb_columns(cols, stride, r);
    fe own;
    base_mult_wide(own, g_wide, cols, stride, R);
    encode_point(encR, own);
    load32(pkw, priv, 2 * e + 1);
    ed_sign_s(s, encR, pkw, msg, len, a, r);
    if (R.is0) {
        store32(sig, 2 * e, encR);
        store32(sig, 2 * e + 1, s);
    }
}

// curve25519_dh_CalculatePublicKey_fast (curve25519_dh.c:162-189) for element e: S = clamp(sk) * B on the Edwards side,
// u = (Z + Y) / (Z - Y); the clamped key written back
C25519_DEV void public_fast_element(void* pk, void* sk, size_t e, const u32* __restrict__ g_wide, unsigned short* cols, int stride)
{
    const Roles R = roles();
    u32 k[8], w[8];
    load32(k, sk, e);
    clamp_words(k);
    if (R.is0) store32(sk, e, k);
    wb_columns(cols, stride, k);
    fe own, Y, Z, t, num, den, zi;
    base_mult_wide(own, g_wide, cols, stride, R);
    fe_qperm<1, 1, 1, 1>(Y, own);
    fe_qperm<3, 3, 3, 3>(Z, own);
    fe_add(t, Z, Y);  fe_carry32(num, t);
    fe_sub(t, Z, Y);  fe_carry32(den, t);
    fe_invert_quad(zi, den);                               // Z = Y (the neutral element) gives 0, like the reference's inversion
    fe_mul(num, num, zi);
    fe_to_words(w, num);
    if (R.is0) store32(pk, e, w);
}

// ed25519_Verify_Check (ed25519_verify.c:287-313) for pair e under ONE key whose wide comb exists (wideP: built for -A, the
// context's row 1): T = s * B + h * (-A) over the two combs, enc(T) compared with the signature's R bytes -- the one-lane
// kernel's steps (k_ed25519_verify_check_wide + the shared inversion) in one launch of quads.  An even h walks h + 1 and one -A
// comes off again (never h + L: a key may carry torsion); s + L when even (L * B = O).
C25519_DEV void verify_check_wide_element(int* verdict, const void* sig, const u32* __restrict__ ctx, const uint8_t* msg, size_t len, size_t e,
                                          const u32* __restrict__ wideB, const u32* __restrict__ wideP, unsigned short* cs, unsigned short* ch,
                                          int stride)
{
    const Roles R = roles();
    u32 pkw[8], Sw[8], h[8], Rw[8], enc[8];
#pragma unroll
    for (int j = 0; j < 8; j++) pkw[j] = ctx[j];
    load32(Rw, sig, 2 * e);
    ed_hram(h, Rw, pkw, msg, len);
    sc_mod(h);
    load32(Sw, sig, 2 * e + 1);                            // raw 256 bits: no s < L check (ed25519_verify.c:308)
    wb_columns(cs, stride, Sw);
    const u32 h_even = wb_columns<false>(ch, stride, h);
    fe own, mult, neutral;
    double_base_mult_wide(own, wideB, cs, wideP, ch, stride, R);
    // - (-A) if h was even, + O otherwise: one more addition either way.  -A = row 1 of the context, (Y+X | Y-X | 2dT | 2Z = 2);
    // the neutral element's row is (1 | 1 | 0 | 2)
    field_raw rf;
    row_field_fetch(rf, ctx + 8 + 32, 0xffffffffu);
    row_field_unpack(mult, rf, R);
    fe_set_u32(neutral, 0);
    neutral.v[0] = ((R.is0 | R.is1) & 1u) | (R.is3 & 2u);
    q_sel(mult, h_even, mult, neutral);
    ge_add_fields(own, mult, R);
    encode_point(enc, own);
    u32 diff = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) diff |= enc[j] ^ Rw[j];
    if (R.is0) verdict[e] = diff == 0 ? 1 : 0;
}

}  // namespace quad
}  // namespace c25519
