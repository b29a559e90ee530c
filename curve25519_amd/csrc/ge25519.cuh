// curve25519_amd/csrc/ge25519.cuh -- twisted-Edwards (a = -1) point arithmetic in extended coordinates,
// one point per lane, and the fixed-base walks built on it.
//
// Device replacement for the reference's L1/L2 Edwards layer:
//   edp_DoublePoint (source/ed25519_sign.c:122), edp_AddAffinePoint (:97), edp_AddPoint
//   (source/ed25519_verify.c:142), edp_ExtPoint2PE (ed25519_sign.c:270), edp_BasePointMult (:215),
//   ed25519_CalculateX (ed25519_verify.c:66), the q_table build of ed25519_Verify_Init (:179-232) and
//   edp_PolyPointMultiply (:243-280).
// The formulas are the reference's (Hisil et al. 2008/522); where the unsigned-limb bound contract of
// fe25519.cuh needs it, a sub-expression is negated consistently ((X:Y:Z:T) and (-X:-Y:-Z:-T) are the
// same point) or folded into a squaring's carry chain.  Outputs leave through a field inversion and a
// canonical encoding, so they are byte-identical to the reference.
#pragma once
#include "curve_constants.cuh"
#include "fe25519.cuh"
#include "sc25519.cuh"

namespace c25519 {

struct ge_ext { fe X, Y, Z, T; };                 // x = X/Z, y = Y/Z, T = XY/Z; all four reduced
struct ge_pa  { fe ypx, ymx, t2d; };              // affine precomputed: y+x, y-x, 2d*x*y   (PA_POINT)
struct ge_pe  { fe ypx, ymx, t2d, z2; };          // projective precomputed, + 2Z               (PE_POINT)

constexpr int PA_WORDS = 30;                      // limbs per table row
constexpr int PE_WORDS = 40;

// p = 2p.  In: X, Y, Z reduced.  Out: all reduced.  4S + 4M.
//   reference: A=X^2 B=Y^2 C=2Z^2 D=-A H=D-B G=D+B F=G-C E=(X+Y)^2+H ; X3=EF Y3=HG Z3=GF T3=EH
//   here:      Hn=-H=A+B, Fn=-F=2Z^2+A-B  (both negated -> all four outputs negated: same point)
// NEED_T = false skips T3 = E*H (one multiplication): T is only read by additions, so inside a run of
// consecutive doublings (the 3 x 64 doublings of the verify table build) only the last one needs it.
template <bool NEED_T = true>
C25519_DEV void ge_double(ge_ext& p)
{
    fe A, B, Hn, G, E, Fn, t;
    fe_sqr(A, p.X);
    fe_sqr(B, p.Y);
    fe_add(Hn, A, B);                    // beta 2
    fe_sub(G, B, A);                     // beta 3
    fe_add(t, p.X, p.Y);                 // beta 2
    fe_sqr_sub(E, t, Hn);                // (X+Y)^2 - A - B, reduced
    fe_sqr2_add_sub(Fn, p.Z, A, B);      // 2Z^2 + A - B, reduced
    fe_mul(p.X, E, Fn);
    fe_mul(p.Y, G, Hn);
    fe_mul(p.Z, G, Fn);
    if (NEED_T) fe_mul(p.T, E, Hn);
}

// p = p + q, q affine precomputed with reduced limbs.  7M.   (edp_AddAffinePoint)
// NEED_T = false skips T3 = E*H: a doubling (or the final affine conversion) that follows never reads T.
template <bool NEED_T = true>
C25519_DEV void ge_add_pa(ge_ext& p, const ge_pa& q)
{
    fe a, b, c, d, e, f, g, h;
    fe_sub(a, p.Y, p.X);                 // beta 3
    fe_mul(a, a, q.ymx);
    fe_add(b, p.Y, p.X);                 // beta 2
    fe_mul(b, b, q.ypx);
    fe_mul(c, p.T, q.t2d);
    fe_add(d, p.Z, p.Z);                 // beta 2
    fe_sub(e, b, a);                     // E = B-A   beta 3
    fe_add(h, b, a);                     // H = B+A   beta 2
    fe_sub(f, d, c);                     // F = D-C   beta 4
    fe_add(g, d, c);                     // G = D+C   beta 3
    fe_mul(p.X, f, e);
    fe_mul(p.Y, g, h);
    if (NEED_T) fe_mul(p.T, e, h);
    fe_mul(p.Z, f, g);
}

// the same addition with T produced on a run-time (wave-uniform) request: ONE copy of the addition where the template
// would put two into a loop (ge_base_mult's last table: T only before the blinding point is added)
C25519_DEV void ge_add_pa_rt(ge_ext& p, const ge_pa& q, bool need_t)
{
    fe a, b, c, d, e, f, g, h;
    fe_sub(a, p.Y, p.X);
    fe_mul(a, a, q.ymx);
    fe_add(b, p.Y, p.X);
    fe_mul(b, b, q.ypx);
    fe_mul(c, p.T, q.t2d);
    fe_add(d, p.Z, p.Z);
    fe_sub(e, b, a);
    fe_add(h, b, a);
    fe_sub(f, d, c);
    fe_add(g, d, c);
    fe_mul(p.X, f, e);
    fe_mul(p.Y, g, h);
    if (need_t) fe_mul(p.T, e, h);
    fe_mul(p.Z, f, g);
}

// r = p + q, q projective precomputed with reduced limbs.  8M.   (edp_AddPoint)
template <bool NEED_T = true>
C25519_DEV void ge_add_pe(ge_ext& r, const ge_ext& p, const ge_pe& q)
{
    fe a, b, c, d, e, f, g, h;
    fe_sub(a, p.Y, p.X);
    fe_mul(a, a, q.ymx);
    fe_add(b, p.Y, p.X);
    fe_mul(b, b, q.ypx);
    fe_mul(c, p.T, q.t2d);
    fe_mul(d, p.Z, q.z2);
    fe_sub(e, b, a);                     // beta 3
    fe_add(h, b, a);                     // beta 2
    fe_sub(f, d, c);                     // beta 3
    fe_add(g, d, c);                     // beta 2
    fe_mul(r.X, e, f);
    fe_mul(r.Y, g, h);
    if (NEED_T) fe_mul(r.T, e, h);
    fe_mul(r.Z, f, g);
}

// r = precomputed form of p, every limb reduced (so table entries can feed fe_sub / fe_mul freely)
C25519_DEV void ge_to_pe(ge_pe& r, const ge_ext& p)
{
    fe t;
    fe_add(t, p.Y, p.X);  fe_carry32(r.ypx, t);
    fe_sub(t, p.Y, p.X);  fe_carry32(r.ymx, t);
    fe_mul(r.t2d, p.T, fe_const(K_2D));
    fe_add(t, p.Z, p.Z);  fe_carry32(r.z2, t);
}

// extended point from a precomputed row: (2x, 2y, 2z, 2xy) -- ed25519_sign.c:226-230 / ed25519_verify.c:258-262
// zr = nullptr: Z = 2 (R = 1; the reference's R is output-neutral).  With a blinding context the starting point is
// spread over the projective class by the context's random R exactly as the reference does (ed25519_sign.c:232-237).
C25519_DEV void ge_from_pa(ge_ext& s, const ge_pa& q, const fe* zr = nullptr)
{
    fe t;
    fe_sub(t, q.ypx, q.ymx);  fe_carry32(s.X, t);
    fe_add(t, q.ypx, q.ymx);  fe_carry32(s.Y, t);
    fe_mul(s.T, q.t2d, fe_const(K_DI));
    if (!zr) {
        fe_set_u32(s.Z, 2);
        return;
    }
    fe_add(t, *zr, *zr);  fe_carry32(s.Z, t);      // Z = 2R
    fe_mul(s.X, s.X, *zr);                         // X = 2xR
    fe_mul(s.T, s.T, *zr);                         // T = 2xyR
    fe_mul(s.Y, s.Y, *zr);                         // Y = 2yR
}

C25519_DEV void ge_from_pe(ge_ext& s, const ge_pe& q)
{
    fe t;
    fe_sub(t, q.ypx, q.ymx);  fe_carry32(s.X, t);
    fe_add(t, q.ypx, q.ymx);  fe_carry32(s.Y, t);
    fe_mul(s.T, q.t2d, fe_const(K_DI));
    s.Z = q.z2;
}

// x from y with the requested parity: sqrt((y^2-1)/(d y^2+1)); like the reference there is NO
// on-curve rejection -- a non-square input just yields whatever the formula yields.   (ed25519_CalculateX)
// Returns all-ones iff the result satisfies the curve equation (v x^2 == u): the reference never looks at this; the
// fast verification path (verify_fast.cuh) does, and only to choose between itself and the reference-order path.
C25519_DEV u32 ge_calc_x_checked(fe& X, const fe& Y, u32 parity)
{
    fe u, v, a, b, t;
    fe one;
    fe_set_u32(one, 1);
    {
        fe_sqr(u, Y);
        fe_mul(v, u, fe_const(K_D));
        fe_sub(t, u, one);  fe_carry32(u, t);            // u = y^2 - 1, reduced
        v.v[0] += 1;                                     // v = d y^2 + 1
        fe_sqr(b, v);
        fe_mul(a, u, b);
        fe_mul(a, a, v);                                 // a = u v^3
        fe_sqr(b, b);                                    // v^4
        fe_mul(b, a, b);                                 // u v^7
    }
    fe_pow2523(b, b);                                    // only y and a live across the 251 squarings: u and v are
    fe_mul(X, b, a);                                     // rebuilt below (1 S + 1 M) instead of holding twenty registers

    fe_sqr(u, Y);
    fe_mul(v, u, fe_const(K_D));
    fe_sub(t, u, one);  fe_carry32(u, t);
    v.v[0] += 1;
    fe_sqr(b, X);                                    // is v x^2 == u ?
    fe_mul(b, b, v);                                 // c = v x^2, reduced
    fe_add(a, b, u);                                 // c + u: zero iff x*sqrt(-1) is the root
    fe_sub(b, b, u);
    u32 bw[8], aw[8];
    fe_to_words(bw, b);
    fe_to_words(aw, a);
    u32 nz = 0, nz2 = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) { nz |= bw[i]; nz2 |= aw[i]; }
    fe_mul(t, X, fe_const(K_SQRTM1));
    fe_select(X, nz ? 0xffffffffu : 0u, t, X);       // :92-93

    u32 xw[8];
    fe_to_words(xw, X);                              // canonical, to read the parity (:95-99)
    fe_neg(t, X);
    fe_select(t, ((xw[0] ^ parity) & 1u) ? 0xffffffffu : 0u, t, X);
    fe_carry32(X, t);
    return (nz == 0 || nz2 == 0) ? 0xffffffffu : 0u;
}

C25519_DEV void ge_calc_x(fe& X, const fe& Y, u32 parity) { (void)ge_calc_x_checked(X, Y, parity); }

// ---- 8-fold base table, staged in LDS ------------------------------------------------------------
// LDS layout is limb-major: word w of row k sits at tbl[w * 256 + k], so the 64 secret row indices of
// a wave spread over the banks instead of marching down one 96-byte row.
C25519_DEV void lds_load_pa(ge_pa& q, const u32* tbl, u32 row)
{
#pragma unroll
    for (int i = 0; i < 10; i++) {
        q.ypx.v[i] = tbl[(i) * 256 + row];
        q.ymx.v[i] = tbl[(10 + i) * 256 + row];
        q.t2d.v[i] = tbl[(20 + i) * 256 + row];
    }
}

// cooperative copy of `words` table words (device resident, a multiple of 4) into this workgroup's LDS
C25519_DEV void lds_stage_words(u32* lds_tbl, const u32* __restrict__ g_tbl, int words)
{
    const uint4* src = reinterpret_cast<const uint4*>(g_tbl);
    uint4* dst = reinterpret_cast<uint4*>(lds_tbl);
    for (int i = threadIdx.x; i < words / 4; i += blockDim.x) dst[i] = src[i];
    __syncthreads();
}

// 8-fold column n of scalar k (not consumed): bit j = bit (31 - n) of word j
C25519_DEV u32 fold8_at(const u32 (&k)[8], int n)
{
    u32 idx = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) idx |= ((k[j] >> (31 - n)) & 1u) << j;
    return idx;
}

// S = k * B.  The reference's walk (edp_BasePointMult, ed25519_sign.c:215-244) is S = T[c0]; S = 2S + T[cn],
// n = 1..31 over one 8-fold table: 31 doublings, 31 additions.  Two regroupings of the same sum give the same point,
// hence the same canonical bytes after the inversion:
//  * BASE_NT tables T_t = 2^((BASE_NT-1-t)*step) * T, step = 32/BASE_NT:
//        sum_n 2^(31-n) T[c_n]  =  sum_{m<step} 2^(step-1-m) * sum_{t<BASE_NT} T_t[c_(t*step+m)]
//    i.e. step-1 doublings instead of 31;
//  * signed digits: k is made odd (k + L when it is even: L*B = O) and written with ALL digits +-1,
//    k' = sum_i s_i 2^i, s_i = 2 w_i - 1 with w = (k' >> 1) | 2^255.  A column of eight teeth is then +-(2^224 + sum of
//    seven +-2^(32 j)) B: 128 rows and a sign instead of 256 rows, so EIGHT tables fit the 120 KiB of LDS that four
//    unsigned ones took, and the walk needs 3 doublings instead of 7.  Rows are negated on the way out of LDS
//    ((y+x, y-x, 2dxy) -> (y-x, y+x, -2dxy)).
// lds_tbl holds T_0 .. T_(BASE_NT-1), limb-major each ([30][BASE_ROWS]); a 1024-thread workgroup (BM_BLOCK) shares it.
// The reference-format table T itself (256 rows: verification's sigma columns, the table test hook) follows the signed
// tables in device memory at REF_TBL_OFFSET.
constexpr int BASE_NT = 8;
constexpr int BASE_STEP = 32 / BASE_NT;
constexpr int BASE_ROWS = 128;
constexpr int BASE_TBL_WORDS = PA_WORDS * BASE_ROWS;
constexpr int REF_TBL_WORDS = PA_WORDS * 256;
constexpr int REF_TBL_OFFSET = BASE_NT * BASE_TBL_WORDS;
// Verification's walk adds sigma*B as ONE signed comb of SC_TEETH teeth SC_COLS bits apart (verify_fast.cuh): 2^(teeth-1)
// rows and a sign, SC_COLS additions riding the walk's last SC_COLS doublings.  The reference's 8-fold table (256 rows,
// 32 additions: edp_PolyPointMultiply, ed25519_verify.c:266-279) is the 8-tooth unsigned member of the family; ten signed
// teeth are 512 rows = 60 KiB of LDS, which two 256-lane workgroups per CU still hold, and 26 additions.
#ifndef C25519_WALK_COMB_TEETH
#define C25519_WALK_COMB_TEETH 10
#endif
constexpr int SC_TEETH = C25519_WALK_COMB_TEETH;
constexpr int SC_COLS = (254 + SC_TEETH - 1) / SC_TEETH;     // the recoded scalar has SC_TEETH * SC_COLS >= 254 digits
constexpr int SC_ROWS = 1 << (SC_TEETH - 1);
constexpr int SC_TBL_WORDS = PA_WORDS * SC_ROWS;
constexpr int SC_TBL_OFFSET = REF_TBL_OFFSET + REF_TBL_WORDS;
constexpr int SC_ROUNDS = (SC_COLS + 3) / 4;                 // digit rounds (four doublings each) that carry columns
constexpr int SIGMA_WORDS = 2 * SC_ROUNDS;                   // two 16-bit columns per word, two words per round
constexpr int ALL_TBL_WORDS = SC_TBL_OFFSET + SC_TBL_WORDS;

// w of the recoding above.  k < 2^255 + 2^254 (clamped scalars, scalars mod L).
C25519_DEV void sc_signed_comb(u32 (&w)[8], const u32 (&k)[8])
{
    const u32 even = (k[0] & 1u) - 1u;                       // all-ones when k is even
    u32 t[8];
    u64 c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        c += (u64)k[i] + (K_L[i] & even);
        t[i] = (u32)c;
        c >>= 32;
    }
#pragma unroll
    for (int i = 0; i < 7; i++) w[i] = (t[i] >> 1) | (t[i + 1] << 31);
    w[7] = (t[7] >> 1) | 0x80000000u;
}

// the row a column byte c selects: tooth 7 is the sign (w bit 0 = digit -1: the whole column is negated), teeth 0..6 the
// index, complemented for a negative column
C25519_DEV void lds_load_pa_signed(ge_pa& q, const u32* tbl, u32 c)
{
    const u32 neg = ((c >> 7) & 1u) - 1u;                     // all-ones: negative column
    const u32 row = (c ^ neg) & 127u;
    const u32* p_ypx = tbl + (neg ? 10 * BASE_ROWS : 0) + row;
    const u32* p_ymx = tbl + (neg ? 0 : 10 * BASE_ROWS) + row;
    fe t, n;
#pragma unroll
    for (int i = 0; i < 10; i++) {
        q.ypx.v[i] = p_ypx[i * BASE_ROWS];
        q.ymx.v[i] = p_ymx[i * BASE_ROWS];
        t.v[i] = tbl[(20 + i) * BASE_ROWS + row];
    }
    fe_neg(n, t);                                            // 2p - 2dxy: beta 2, fine as the second factor of a product
    fe_select(q.t2d, neg, n, t);
}

// FINAL_T: also produce T of the result (needed when another addition follows, i.e. the blinding point).
template <bool FINAL_T = false>
C25519_DEV void ge_base_mult(ge_ext& S, const u32 (&k)[8], const u32* lds_tbl, const fe* zr = nullptr)
{
    u32 w[8];
    sc_signed_comb(w, k);
    ge_pa q;
    lds_load_pa_signed(q, lds_tbl, fold8_at(w, 0));
    ge_from_pa(S, q, zr);
#pragma unroll 1
    for (int m = 0; m < BASE_STEP; m++) {
        if (m) ge_double(S);
        // tables 0 .. BASE_NT-2 (the first one is the starting point when m == 0): T feeds the next addition.
        // Kept as a loop: one copy of the addition in the instruction cache instead of BASE_NT.
#pragma unroll 1
        for (int t = m ? 0 : 1; t < BASE_NT - 1; t++) {
            lds_load_pa_signed(q, lds_tbl + t * BASE_TBL_WORDS, fold8_at(w, t * BASE_STEP + m));
            C25519_SCHED_FENCE();          // the row is complete before the addition starts: without it the 1024-thread
            ge_add_pa<true>(S, q);         // kernels (128 registers) spilled two to four registers
        }
        // last table: a doubling or the affine conversion follows, neither reads T
        lds_load_pa_signed(q, lds_tbl + (BASE_NT - 1) * BASE_TBL_WORDS, fold8_at(w, (BASE_NT - 1) * BASE_STEP + m));
        C25519_SCHED_FENCE();
        if (FINAL_T) ge_add_pa_rt(S, q, m == BASE_STEP - 1);
        else ge_add_pa<false>(S, q);
    }
}

// ---- the WIDE fixed-base comb, read through L2 (tunable BASE_COMB = 1) ----------------------------------------------
// The LDS comb above is capped at 8 teeth by 120 KiB of LDS: 31 additions + 3 doublings per scalar.  A signed comb of
// WB_TEETH = 13 teeth WB_COLS = 20 bits apart, cut into WB_NT = 4 tables T_t = 2^((WB_NT-1-t) * WB_STEP) * T of 4096 rows
// each, needs 19 additions + 4 doublings -- a third fewer field products -- at 4 x 4096 rows x 128 bytes = 2 MiB of table:
// not LDS, but resident in every XCD's 4 MB L2.  A lane fetches the row its (secret) column selects straight from
// global memory, exactly as the reference indexes its table (ed25519_sign.c:239-243): one 128-byte line per row (packed
// canonical Y+X | Y-X | 2dT | pad), no staging, no LDS table -- so the workgroup shape is free (256 lanes, four waves per
// SIMD: while one wave waits ~500 cycles for its row, the other three issue MADs).  The columns are gathered ONCE, with
// compile-time bit positions, into 20 x 16 bits per lane parked in LDS (the walk's loops stay rolled: a run-time column
// number would index the scalar's registers dynamically).
// Same recoding as above with 260 digits: w = (k' >> 1) | 2^259; column p = bits p, p + 20, ..., p + 240 of w, the top one
// its sign; sum_p 2^p T[col_p] = sum_{m < STEP} 2^(STEP-1-m) sum_{t < NT} T_t[col_(COLS-1-(t*STEP+m))].
constexpr int WB_TEETH = 13;
constexpr int WB_COLS = 20;
constexpr int WB_NT = 4;
constexpr int WB_STEP = WB_COLS / WB_NT;
constexpr int WB_ROWS = 1 << (WB_TEETH - 1);
constexpr int WB_ROW_WORDS = 32;                             // one 128-byte line: 3 x 8 words + 8 of padding
constexpr size_t WB_TBL_WORDS = (size_t)WB_NT * WB_ROWS * WB_ROW_WORDS;    // 2 MiB
static_assert(WB_NT * WB_STEP == WB_COLS && WB_TEETH * WB_COLS >= 256, "the wide comb covers 256 + bits");

// the 20 columns of k in the order the walk consumes them (s = m * WB_NT + t), 16 bits each, to cols[s * stride].
// The all-digits-+-1 form needs an ODD value.  ADD_L: an even k becomes k + L -- the same multiple of any point of order L
// (the base point).  !ADD_L, for a point that may carry a torsion component (a public key: L * P != O): an even k becomes
// k + 1 and the caller takes one P off again (returns all-ones in that case).
template <bool ADD_L = true, typename ColT>
C25519_DEV u32 wb_columns(ColT* cols, int stride, const u32 (&k)[8])
{
    const u32 even = (k[0] & 1u) - 1u;                       // all-ones when k is even
    u32 t[9], w[9];
    u64 c = ADD_L ? 0 : (even & 1u);
#pragma unroll
    for (int i = 0; i < 8; i++) {
        c += (u64)k[i] + (ADD_L ? (K_L[i] & even) : 0u);
        t[i] = (u32)c;
        c >>= 32;
    }
    t[8] = (u32)c;                                           // k + L < 2^256 + 2^253: one more bit at most
#pragma unroll
    for (int i = 0; i < 8; i++) w[i] = (t[i] >> 1) | (t[i + 1] << 31);
    w[8] = t[8] >> 1;
    constexpr int TOP = WB_TEETH * WB_COLS - 1;              // digit TOP is +1
    w[TOP >> 5] |= 1u << (TOP & 31);
#pragma unroll
    for (int m = 0; m < WB_STEP; m++)
#pragma unroll
        for (int tt = 0; tt < WB_NT; tt++) {
            const int p = WB_COLS - 1 - (tt * WB_STEP + m);
            u32 idx = 0;
#pragma unroll
            for (int j = 0; j < WB_TEETH; j++) {
                const int bit = WB_COLS * j + p;
                idx |= ((w[bit >> 5] >> (bit & 31)) & 1u) << j;
            }
            cols[(m * WB_NT + tt) * stride] = (ColT)idx;
        }
    return even;
}

// the row a 13-bit column c selects in table `tbl` (4096 packed rows): tooth 12 is the sign
C25519_DEV void wb_load_pa_signed(ge_pa& q, const u32* __restrict__ tbl, u32 c)
{
    const u32 neg = ((c >> (WB_TEETH - 1)) & 1u) - 1u;        // all-ones: negative column
    const u32 row = (c ^ neg) & (u32)(WB_ROWS - 1);
    const uint4* r = reinterpret_cast<const uint4*>(tbl + (size_t)row * WB_ROW_WORDS);
    const uint4* p_ypx = r + (neg ? 2 : 0);                  // a negative column swaps Y+X and Y-X ...
    const uint4* p_ymx = r + (neg ? 0 : 2);
    const uint4 a0 = p_ypx[0], a1 = p_ypx[1], b0 = p_ymx[0], b1 = p_ymx[1], c0 = r[4], c1 = r[5];
    const u32 wa[8] = { a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w };
    const u32 wb[8] = { b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w };
    const u32 wc[8] = { c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w };
    fe t, n;
    fe_from_words(q.ypx, wa);
    fe_from_words(q.ymx, wb);
    fe_from_words(t, wc);
    fe_neg(n, t);                                            // ... and negates 2dxy
    fe_select(q.t2d, neg, n, t);
}

// S = k * B over the wide comb.  g_wide: the WB_NT packed tables in device memory; cols: this lane's parked columns
// (wb_columns), `stride` elements apart.  FINAL_T as in ge_base_mult.
template <bool FINAL_T = false, typename ColT>
C25519_DEV void ge_base_mult_wide(ge_ext& S, const u32* __restrict__ g_wide, const ColT* cols, int stride, const fe* zr = nullptr)
{
    ge_pa q;
    wb_load_pa_signed(q, g_wide, cols[0]);
    ge_from_pa(S, q, zr);
#pragma unroll 1
    for (int m = 0; m < WB_STEP; m++) {
        if (m) ge_double(S);
#pragma unroll 1
        for (int t = m ? 0 : 1; t < WB_NT - 1; t++) {
            wb_load_pa_signed(q, g_wide + (size_t)t * WB_ROWS * WB_ROW_WORDS, cols[(m * WB_NT + t) * stride]);
            C25519_SCHED_FENCE();
            ge_add_pa<true>(S, q);
        }
        wb_load_pa_signed(q, g_wide + (size_t)(WB_NT - 1) * WB_ROWS * WB_ROW_WORDS, cols[(m * WB_NT + WB_NT - 1) * stride]);
        C25519_SCHED_FENCE();
        if (FINAL_T) ge_add_pa_rt(S, q, m == WB_STEP - 1);
        else ge_add_pa<false>(S, q);
    }
}

// S = s * B + h * P over TWO wide combs walked together -- the base point's and one built for a point P (the two-phase
// verification's key, -A: engine.hip, k_ed25519_verify_check_wide): 39 additions and the same 4 doublings.  colsB / colsP:
// the lane's parked columns of s and h (wb_columns, the latter without "+ L": P may have a torsion component);
// h_was_even: h's columns encode h + 1, so one P comes off at the end (p_words: P's affine precomputed form, fetched then).
template <typename ColT>
C25519_DEV void ge_double_base_mult_wide(ge_ext& S, const u32* __restrict__ wideB, const ColT* colsB, const u32* __restrict__ wideP,
                                         const ColT* colsP, int stride, u32 h_was_even, const u32* __restrict__ p_words /* Y+X | Y-X | 2dT of P, 8 words each */)
{
    ge_pa q;
    wb_load_pa_signed(q, wideB, colsB[0]);
    ge_from_pa(S, q);
    wb_load_pa_signed(q, wideP, colsP[0]);
    C25519_SCHED_FENCE();
    ge_add_pa<true>(S, q);
#pragma unroll 1
    for (int m = 0; m < WB_STEP; m++) {
        if (m) ge_double(S);
#pragma unroll 1
        for (int t = m ? 0 : 1; t < WB_NT; t++) {
            const size_t off = (size_t)t * WB_ROWS * WB_ROW_WORDS;
            wb_load_pa_signed(q, wideB + off, colsB[(m * WB_NT + t) * stride]);
            C25519_SCHED_FENCE();
            ge_add_pa<true>(S, q);
            wb_load_pa_signed(q, wideP + off, colsP[(m * WB_NT + t) * stride]);
            C25519_SCHED_FENCE();
            ge_add_pa_rt(S, q, t != WB_NT - 1 || m == WB_STEP - 1);   // T where another addition follows (a doubling does not read it)
        }
    }
    // - P if h was even, + O otherwise: one more addition either way (the neutral element's row is (1, 1, 0))
    fe one, zero, n, f;
    u32 w[8];
    fe_set_u32(one, 1);
    fe_set_u32(zero, 0);
#pragma unroll
    for (int j = 0; j < 8; j++) w[j] = p_words[8 + j];
    fe_from_words(f, w);
    fe_select(q.ypx, h_was_even, f, one);                    // -P swaps Y+X and Y-X ...
#pragma unroll
    for (int j = 0; j < 8; j++) w[j] = p_words[j];
    fe_from_words(f, w);
    fe_select(q.ymx, h_was_even, f, one);
#pragma unroll
    for (int j = 0; j < 8; j++) w[j] = p_words[16 + j];
    fe_from_words(f, w);
    fe_neg(n, f);                                            // ... and negates 2dxy
    fe_select(q.t2d, h_was_even, n, zero);
    C25519_SCHED_FENCE();
    ge_add_pa<false>(S, q);
}

// affine canonical words of S: x = X/Z, y = Y/Z   (tail of edp_BasePointMultiply, ed25519_sign.c:265-267)
C25519_DEV void ge_to_affine_words(u32 (&xw)[8], u32 (&yw)[8], const ge_ext& S)
{
    fe zi, t;
    fe_invert(zi, S.Z);
    fe_mul(t, S.X, zi);  fe_to_words(xw, t);
    fe_mul(t, S.Y, zi);  fe_to_words(yw, t);
}

// ed25519_PackPoint / ecp_EncodeInt (curve25519_utils.c:77-98): y with the parity of x in bit 255
C25519_DEV void ge_pack(u32 (&out)[8], const u32 (&xw)[8], const u32 (&yw)[8])
{
#pragma unroll
    for (int i = 0; i < 8; i++) out[i] = yw[i];
    out[7] = (yw[7] & 0x7fffffffu) | (xw[0] << 31);
}

// ---- 4-fold tables (verify) ------------------------------------------------------------------------------
// 16 rows of PE points, q_table[k] = sum over set bits i of k of 2^(64 i) * Q   (ed25519_Verify_Init :199-229).
// Two storage formats behind the same load/store interface:
//   QTableLimbs -- lane-private rows of 40 limbs (160 B) in a global scratch slab, rows of one lane contiguous
//                  (2560 B per lane).  A row lookup by a secret index touches 3 cache lines of that lane's
//                  own memory; no conversion work.  Used by the one-shot ed25519_VerifySignature path.
//   QTableCanon -- the reference's EDP_SIGV_CTX row layout: 4 field elements of 8 canonical words (128 B per
//                  row, 2048 B per table).  This is what ed25519_Verify_Init hands back to the caller
//                  (fits the reference's 2080-byte context) and what Verify_Check consumes.
constexpr size_t QTABLE_LIMB_WORDS = 16 * PE_WORDS;       // per lane
constexpr size_t QTABLE_CANON_WORDS = 16 * 32;            // per table

struct QTableLimbs {
    u32* base;                                             // this lane's 16 rows
    C25519_DEV void store(int e, const ge_pe& q) const
    {
        uint4* row = reinterpret_cast<uint4*>(base + (size_t)e * PE_WORDS);
        const fe* f[4] = { &q.ypx, &q.ymx, &q.t2d, &q.z2 };
        u32 w[PE_WORDS];
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int i = 0; i < 10; i++) w[10 * j + i] = f[j]->v[i];
#pragma unroll
        for (int g = 0; g < 10; g++) row[g] = make_uint4(w[4 * g], w[4 * g + 1], w[4 * g + 2], w[4 * g + 3]);
    }
    C25519_DEV void load(ge_pe& q, u32 e) const
    {
        const uint4* row = reinterpret_cast<const uint4*>(base + (size_t)e * PE_WORDS);
        u32 w[PE_WORDS];
#pragma unroll
        for (int g = 0; g < 10; g++) {
            const uint4 v = row[g];
            w[4 * g] = v.x; w[4 * g + 1] = v.y; w[4 * g + 2] = v.z; w[4 * g + 3] = v.w;
        }
#pragma unroll
        for (int i = 0; i < 10; i++) {
            q.ypx.v[i] = w[i]; q.ymx.v[i] = w[10 + i]; q.t2d.v[i] = w[20 + i]; q.z2.v[i] = w[30 + i];
        }
    }
};

struct QTableCanon {
    u32* base;                                             // 16 rows x (YpX, YmX, T2d, Z2) x 8 words
    C25519_DEV void store(int e, const ge_pe& q) const
    {
        uint4* row = reinterpret_cast<uint4*>(base + (size_t)e * 32);
        const fe* f[4] = { &q.ypx, &q.ymx, &q.t2d, &q.z2 };
#pragma unroll
        for (int j = 0; j < 4; j++) {
            u32 w[8];
            fe_to_words(w, *f[j]);
            row[2 * j] = make_uint4(w[0], w[1], w[2], w[3]);
            row[2 * j + 1] = make_uint4(w[4], w[5], w[6], w[7]);
        }
    }
    C25519_DEV void load(ge_pe& q, u32 e) const
    {
        const uint4* row = reinterpret_cast<const uint4*>(base + (size_t)e * 32);
        fe* f[4] = { &q.ypx, &q.ymx, &q.t2d, &q.z2 };
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint4 a = row[2 * j], b = row[2 * j + 1];
            const u32 w[8] = { a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w };
            fe_from_words(*f[j], w);
        }
    }
};

// fill all 16 rows from Q (= -A).  Q is consumed.
template <typename Tbl>
C25519_DEV void qtable_build(const Tbl& tbl, ge_ext& Q)
{
    ge_pe pe;
    ge_ext T;
    fe_set_u32(pe.ypx, 1); fe_set_u32(pe.ymx, 1); fe_set_u32(pe.t2d, 0); fe_set_u32(pe.z2, 2);
    tbl.store(0, pe);
    ge_to_pe(pe, Q);
    tbl.store(1, pe);

#pragma unroll 1
    for (int blk = 1; blk < 4; blk++) {               // Q <- 2^64 Q, then fill rows [2^blk, 2^(blk+1))
#pragma unroll 1
        for (int i = 0; i < 63; i++) ge_double<false>(Q);
        ge_double<true>(Q);
        const int top = 1 << blk;
        ge_to_pe(pe, Q);
        tbl.store(top, pe);
#pragma unroll 1
        for (int s = 1; s < top; s++) {               // row top+s = Q + row s   (QTABLE_SET, :175-177)
            tbl.load(pe, (u32)s);
            ge_add_pe(T, Q, pe);
            ge_to_pe(pe, T);
            tbl.store(top + s, pe);
        }
    }
}

// S = s*B + h*Q by the interleaved 4-fold / 8-fold walk   (edp_PolyPointMultiply :243-280).
// s and h (8 words each) are consumed.
template <typename Tbl>
C25519_DEV void ge_poly_mult(ge_ext& S, u32 (&s)[8], u32 (&h)[8], const Tbl& tbl, const u32* lds_tbl)
{
    ge_pe pe;
    ge_pa pa;
    tbl.load(pe, fold4_next(h, false));
    ge_from_pe(S, pe);
#pragma unroll 1
    for (int i = 1; i < 32; i++) {
        ge_double(S);
        tbl.load(pe, fold4_next(h, false));
        ge_add_pe<false>(S, S, pe);
    }
#pragma unroll 1
    for (int i = 32; i < 64; i++) {
        ge_double(S);
        lds_load_pa(pa, lds_tbl, fold8_next(s));
        ge_add_pa<true>(S, pa);                // T feeds the addition that follows
        tbl.load(pe, fold4_next(h, true));
        ge_add_pe<false>(S, S, pe);
    }
}

}  // namespace c25519
