// curve25519_amd/csrc/safegcd25519.cuh -- 1/z mod p = 2^255 - 19 by Bernstein-Yang division steps ("safegcd", 2019) in constant
// time, on 256-bit little-endian words.
//
// The reference inverts by Fermat: z^(p-2), 254 squarings and 11 products (ecp_Inverse, source/curve25519_utils.c /
// curve25519_mehdi.c:340-409).  One lane of this library pays ~25 000 instructions for that chain (fe_invert_fermat,
// fe25519.cuh), and wherever an operation runs on a lane, a quad or a wave of its own -- a single call through the reference's
// prototypes, calls of a few thousand elements, the shared inversion's lone waves -- the chain's LENGTH is what the call waits
// for: 44-56 us of a 100-170 us call.  The value 1/z is unique, so any algorithm gives the reference's bytes; this one is the
// 2-adic Euclid of Bernstein and Yang: 600 division steps on (f, g) = (p, z), 30 at a time on the low words only (~20
// instructions a step, no multiplier), the 2 x 2 transition matrix of each batch then applied to the 270-bit f, g and to the
// cofactors d, e (mod p) with 32 x 32 -> 64-bit signed multiply-adds: ~16 000 instructions, no secret-dependent branch or address.
// Layout and invariants are those of the signed-30-bit-limb formulation proven for libsecp256k1's modinv32 (nine limbs, d and e
// kept in (-2p, p), 20 batches of 30 steps from zeta = -1, whose bound of 590 steps for 256-bit moduli covers this 255-bit one);
// p's limbs are written signed -- (-19, 0, ..., 0, 2^15) -- so the multiple of p that clears a batch's low 30 bits costs two
// multiply-adds, not nine.  z = 0 gives 0, as z^(p-2) does (low-order X25519 inputs come out as zero bytes, curve25519_dh.c:148).
//
// Plain C++ (the compiler emits v_mad_i64_i32 / v_ashrrev_i64): the CPU model runs the same source (tests/host_emul).
#pragma once
// (included by fe25519.cuh behind the primitives: valu_gfx950.cuh on the device, tests/host_emul/valu_model.h in the CPU model)

namespace c25519 {

typedef int32_t i32;
typedef int64_t i64;

constexpr i32 SG_M30 = (i32)0x3fffffff;
constexpr u32 SG_P_INV30 = 0x179435e5u;                  // p^-1 mod 2^30

struct sg30 { i32 v[9]; };                               // sum v[i] 2^(30 i); v[0..7] in [0, 2^30), v[8] signed
struct sg_mat { i32 u, v, q, r; };                       // one batch's transition matrix, scaled by 2^30

// 30 division steps on the low words of f (odd) and g: zeta' and the matrix t with  2^30 (f', g') = t (f, g).
// The steps run in the UNSHIFTED form: at step i the pair (X, Y) is (f, g) 2^i, (u, q) or (v, r) -- "g odd" is bit i of Y(f, g), g
// is never halved, and every pair runs the same recurrence
//      Y <- Y + (g odd ? (swap ? -X : X) : 0)        X <- 2 (swap ? the old Y : X)          swap: g odd and zeta < 0
// (the halving of g and the doubling of the matrix's f-row are the one doubling of X; all of it mod 2^32, which keeps the low
// 30 bits of f and g and the whole matrix exact); zeta <- -zeta - 2 on a swap, zeta - 1 otherwise.  sg_steps30 (valu_gfx950.cuh;
// the CPU model's is valu_model.h) is the thirty steps' instruction sequence: 17 a step, no secret-dependent branch or address.
C25519_DEV i32 sg_divsteps30(i32 zeta, u32 f, u32 g, sg_mat& t)
{
    u32 u = 1, v = 0, q = 0, r = 1;
    sg_steps30(zeta, f, g, u, q, v, r);
    t.u = (i32)u; t.v = (i32)v; t.q = (i32)q; t.r = (i32)r;
    return zeta;
}

// (f, g) <- t (f, g) / 2^30   (exact: the matrix was made to clear the low 30 bits)
C25519_DEV void sg_update_fg(sg30& f, sg30& g, const sg_mat& t)
{
    i64 cf = mad2_i64_i32(0, t.u, f.v[0], t.v, g.v[0]);
    i64 cg = mad2_i64_i32(0, t.q, f.v[0], t.r, g.v[0]);
    cf >>= 30;  cg >>= 30;
#pragma unroll
    for (int i = 1; i < 9; i++) {
        cf = mad2_i64_i32(cf, t.u, f.v[i], t.v, g.v[i]);
        cg = mad2_i64_i32(cg, t.q, f.v[i], t.r, g.v[i]);
        f.v[i - 1] = (i32)cf & SG_M30;  cf >>= 30;
        g.v[i - 1] = (i32)cg & SG_M30;  cg >>= 30;
    }
    f.v[8] = (i32)cf;
    g.v[8] = (i32)cg;
}

// (d, e) <- (t (d, e) + p (md, me)) / 2^30 with md, me chosen to make the division exact; d, e stay in (-2p, p)
C25519_DEV void sg_update_de(sg30& d, sg30& e, const sg_mat& t)
{
    const i32 sd = d.v[8] >> 31, se = e.v[8] >> 31;      // a negative d (e) adds its matrix column to the multiple of p
    i32 md = (t.u & sd) + (t.v & se);
    i32 me = (t.q & sd) + (t.r & se);
    i64 cd = mad2_i64_i32(0, t.u, d.v[0], t.v, e.v[0]);
    i64 ce = mad2_i64_i32(0, t.q, d.v[0], t.r, e.v[0]);
    md -= (i32)((SG_P_INV30 * (u32)cd + (u32)md) & (u32)SG_M30);
    me -= (i32)((SG_P_INV30 * (u32)ce + (u32)me) & (u32)SG_M30);
    cd = mad_i64_i32(cd, -19, md);                       // p = (-19, 0, ..., 0, 2^15) in signed limbs
    ce = mad_i64_i32(ce, -19, me);
    cd >>= 30;  ce >>= 30;
#pragma unroll
    for (int i = 1; i < 9; i++) {
        cd = mad2_i64_i32(cd, t.u, d.v[i], t.v, e.v[i]);
        ce = mad2_i64_i32(ce, t.q, d.v[i], t.r, e.v[i]);
        if (i == 8) { cd = mad_i64_i32(cd, 1 << 15, md);  ce = mad_i64_i32(ce, 1 << 15, me); }
        d.v[i - 1] = (i32)cd & SG_M30;  cd >>= 30;
        e.v[i - 1] = (i32)ce & SG_M30;  ce >>= 30;
    }
    d.v[8] = (i32)cd;
    e.v[8] = (i32)ce;
}

// d in (-2p, p), negated where sign < 0, to [0, p) with limbs in [0, 2^30)
C25519_DEV void sg_normalize(sg30& d, i32 sign)
{
    const i32 P0 = -19, P8 = 1 << 15;
    i32 add = d.v[8] >> 31;                              // negative: + p
    const i32 neg = sign >> 31;
    d.v[0] += P0 & add;  d.v[8] += P8 & add;
#pragma unroll
    for (int i = 0; i < 9; i++) d.v[i] = (d.v[i] ^ neg) - neg;
#pragma unroll
    for (int i = 0; i < 8; i++) { d.v[i + 1] += d.v[i] >> 30;  d.v[i] &= SG_M30; }
    add = d.v[8] >> 31;                                  // still negative (it was in (-2p, -p], or the negation made it so): + p
    d.v[0] += P0 & add;  d.v[8] += P8 & add;
#pragma unroll
    for (int i = 0; i < 8; i++) { d.v[i + 1] += d.v[i] >> 30;  d.v[i] &= SG_M30; }
}

// 30-bit limbs of 256-bit words (limb i = bits 30 i .. 30 i + 29) and back
C25519_DEV void sg_from_words(sg30& g, const u32 (&in)[8])
{
    g.v[0] = (i32)(in[0] & (u32)SG_M30);
#pragma unroll
    for (int i = 1; i < 8; i++) {
        const int bit = 30 * i, w = bit >> 5, s = bit & 31;          // s = 30, 28, ..., 18: the limb straddles words w, w + 1
        g.v[i] = (i32)(alignbit32(in[w + 1], in[w], s) & (u32)SG_M30);
    }
    g.v[8] = (i32)(in[7] >> 16);
}
C25519_DEV void sg_to_words(u32 (&out)[8], const sg30& d)
{
    out[0] = (u32)d.v[0] | ((u32)d.v[1] << 30);
#pragma unroll
    for (int w = 1; w < 8; w++) {
        const int lo = (32 * w) / 30, s = 32 * w - 30 * lo;          // word w starts s = 2 w bits into limb lo = w: two limbs cover it
        out[w] = ((u32)d.v[lo] >> s) | ((u32)d.v[lo + 1] << (30 - s));
    }
}
C25519_DEV i32 sg_p_limb(int i) { return i == 0 ? SG_M30 - 18 : i == 8 ? (1 << 15) - 1 : SG_M30; }    // p = 2^255 - 19

// out = 1 / in mod p as canonical words; in: canonical words of a value in [0, p).  0 -> 0.
C25519_DEV void sg_invert_words(u32 (&out)[8], const u32 (&in)[8])
{
    sg30 f, g, d, e;
    sg_from_words(g, in);
#pragma unroll
    for (int i = 0; i < 9; i++) { f.v[i] = sg_p_limb(i); d.v[i] = 0; e.v[i] = 0; }
    e.v[0] = 1;
    i32 zeta = -1;
#pragma unroll 1
    for (int it = 0; it < 20; it++) {
        sg_mat t;
        zeta = sg_divsteps30(zeta, (u32)f.v[0] | ((u32)f.v[1] << 30), (u32)g.v[0] | ((u32)g.v[1] << 30), t);
        sg_update_de(d, e, t);
        sg_update_fg(f, g, t);
    }
    // g = 0 and f = +-1 now (+-p for in = 0, where d = 0): 1/in = sign(f) d
    sg_normalize(d, f.v[8]);
    sg_to_words(out, d);
}

// The same value by the FOUR LANES OF AN ALIGNED QUAD (lane & 3; all four active, all holding the same `in`): what an operation
// with a quad (quad25519.cuh) or a wave (coop25519.cuh: the sixteen lanes of a row) to itself runs.  Both halves of a batch split:
//  * the thirty steps: one pair per lane (sg_divsteps30_quad's recurrences; lane 0 (f, g), lane 1 (u, q), lane 2 (v, r));
//  * the matrix application: lane 0 computes the new f, lane 1 the new g, lane 2 the new d, lane 3 the new e -- a lane keeps
//    `own` (the number it computes) and `other` (its partner's: lane ^ 1), its row of the matrix as (a, b) with
//    own' = (a own + b other [+ p md]) / 2^30: (u, v) on the even lanes, (r, q) on the odd ones; the partners then swap their
//    results (nine v_mov_b32_dpp quad_perm:[1,0,3,2]).  The multiple of p that makes the division exact is d's and e's only
//    (`de`: all-ones on lanes 2, 3); the formula for md is symmetric in (own, other) with (a, b).
// 20 multiply-adds and ~75 other instructions a batch instead of 76 and ~100; ~8 300 instructions an inversion against the one
// lane's ~13 700.
template <int CTRL>
C25519_DEV u32 sg_quad_perm(u32 x) { return (u32)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, 0xf, 0xf, true); }

C25519_DEV void sg_invert_words_quad(u32 (&out)[8], const u32 (&in)[8])
{
    const u32 q = threadIdx.x & 3u;
    const u32 lane1 = q == 1 ? 0xffffffffu : 0u, lane2 = q == 2 ? 0xffffffffu : 0u;
    const u32 odd = (u32)0 - (q & 1u), de = (u32)0 - (q >> 1);
    sg30 own, other;
    {
        sg30 z;
        sg_from_words(z, in);
#pragma unroll
        for (int i = 0; i < 9; i++) {                     // lane 0: (p, z)   lane 1: (z, p)   lane 2: (0, 1)   lane 3: (1, 0)
            const u32 zi = (u32)z.v[i], pi = (u32)sg_p_limb(i), one = i == 0 ? 1u : 0u;
            own.v[i] = (i32)((((zi & odd) | (pi & ~odd)) & ~de) | (one & odd & de));
            other.v[i] = (i32)((((pi & odd) | (zi & ~odd)) & ~de) | (one & ~odd & de));
        }
    }
    i32 zeta = -1;
#pragma unroll 1
    for (int it = 0; it < 20; it++) {
        // lane 0's pair is (f, g) = (own, other); lanes 1 and 2 start from the identity's columns; lane 3 idles along
        u32 X = (((u32)own.v[0] | ((u32)own.v[1] << 30)) & ~(lane1 | lane2)) | (lane1 & 1u);
        u32 Y = (((u32)other.v[0] | ((u32)other.v[1] << 30)) & ~(lane1 | lane2)) | (lane2 & 1u);
        sg_steps30_quad(zeta, X, Y);
        // u = X of lane 1, q = Y of lane 1, v = X of lane 2, r = Y of lane 2;  (a, b) = (u, v) on even lanes, (r, q) on odd ones
        const i32 a = (i32)sg_quad_perm<0x99>((X & ~lane2) | (Y & lane2));     // quad_perm:[1,2,1,2]
        const i32 b = (i32)sg_quad_perm<0x66>((Y & ~lane2) | (X & lane2));     // quad_perm:[2,1,2,1]
        const i32 s_own = own.v[8] >> 31, s_other = other.v[8] >> 31;          // a negative d (e) adds its matrix column to the multiple of p
        i32 md = ((a & s_own) + (b & s_other)) & (i32)de;
        i64 c = mad2_i64_i32(0, a, own.v[0], b, other.v[0]);
        md = (md - (i32)((SG_P_INV30 * (u32)c + (u32)md) & (u32)SG_M30)) & (i32)de;
        c = mad_i64_i32(c, -19, md);                      // p = (-19, 0, ..., 0, 2^15) in signed limbs
        c >>= 30;
#pragma unroll
        for (int i = 1; i < 9; i++) {
            c = mad2_i64_i32(c, a, own.v[i], b, other.v[i]);
            if (i == 8) c = mad_i64_i32(c, 1 << 15, md);
            own.v[i - 1] = (i32)c & SG_M30;  c >>= 30;
        }
        own.v[8] = (i32)c;
#pragma unroll
        for (int i = 0; i < 9; i++) other.v[i] = (i32)sg_quad_perm<0xb1>((u32)own.v[i]);   // quad_perm:[1,0,3,2]
    }
    // g = 0 and f = +-1 now (+-p for in = 0, where d = 0): 1/in = sign(f) d -- d is lane 2's own, f lane 0's
    sg30 d;
#pragma unroll
    for (int i = 0; i < 9; i++) d.v[i] = (i32)sg_quad_perm<0xaa>((u32)own.v[i]);           // quad_perm:[2,2,2,2]
    sg_normalize(d, (i32)sg_quad_perm<0x00>((u32)own.v[8]));
    sg_to_words(out, d);
}

}  // namespace c25519
