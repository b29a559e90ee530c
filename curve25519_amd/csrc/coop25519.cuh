// curve25519_amd/csrc/coop25519.cuh -- ONE operation per wave: the latency shape of the path.
//
// The batch kernels give every lane its own operation, so a call of one (the reference's own prototypes,
// include/curve25519_dh.h) waits for one lane to walk the whole 255-step dependent chain alone: 0.75 ms for
// curve25519_dh_CreateSharedKey against 93 us for the reference on one host core (profiles/r03_hostapi.txt).  Here the 64
// lanes of a wave share ONE operation: a field element lives limb-per-lane in a 16-lane row (lane c of the row holds limb
// c, ten of the sixteen lanes used), the four rows compute four field products at the same time, and a product is ten
// v_mad_u64_u32 per lane (its column) instead of a hundred.  A ladder step (curve25519_dh.c:57-84) is three such product
// levels -- {(x1-z1)(x2+z2), (x2-z2)(x1+z1), (x+z)^2, (x-z)^2}, {x3, (..)^2, x4, z4}, {z3 = (..)^2 * xb} -- with the
// additions between them done in registers after a row exchange (v_permlane16_swap / v_permlane32_swap).
//
// Operands travel through LDS (a wave's LDS operations execute in order: no barrier, no s_waitcnt between a lane's
// store and another lane's load): a value is stored once in the two forms a product reads it in,
//   A-form   a[0..9]                        -- the multiplicand, every lane of a row reads all ten (broadcast reads)
//   Y-forms  yo[s], ye[s], s = 0..19        -- the multiplier b pre-scaled and laid out so that column k's ten terms
//            yo = [19 b_0 .. 19 b_9 | b_0 .. b_9],  ye = the same with the odd limbs doubled
//            are the ten CONSECUTIVE entries s = k+1 .. k+10: term t pairs a_(9-t) with yo[k+1+t] (t odd: a's limb is
//            even) or ye[k+1+t] (t even: a's limb is odd, so an odd b limb takes the factor 2 of radix 2^25.5); entries
//            below 10 are the wrapped ones (2^255 = 19).
// Column sums are carried across lanes with DPP row shifts: S = l0 + 2^w l1 + 2^51 l2, limb_c = l0_c + l1_(c-1) + l2_(c-2)
// (lanes 9 / 8,9 wrap into lanes 0 / 0,1 times 19), then one more single-bit pass -- the result is "reduced" in the sense
// of fe25519.cuh's bound contract, so the formulas' bounds are those tools/fe_bounds.py checks for the batch kernels.
//
// Side channels: no secret-dependent branch anywhere, and in the LADDER no secret-dependent address either (the per-bit choice
// is a v_cndmask on register values).  The FIXED-BASE walk (sign, key pair, public_fast) fetches table rows from global memory
// by the secret comb column, exactly as the reference indexes its table (ed25519_sign.c:239-243) and as the batch kernels'
// wide comb does: which cache lines of the 2 MiB table a call touches depends on the nonce / key (the LDS comb of the batch
// kernels shows bank conflicts only) -- the reference's own exposure, stated in INTEGRATION.md; a blinding context runs the batch
// kernels.  The secret-derived LDS contents (operand slots, the fetched rows) are wiped before a kernel returns (wipe()).
// Results are the reference's bytes (the final inversion, multiplication and canonical encoding are the batch kernels' own
// code, run by every lane redundantly).
#pragma once
#include "fe25519.cuh"
#include "ge25519.cuh"
#include "verify_fast.cuh"

namespace c25519 {
namespace coop {

// LDS words of one value in operand form: A-form at 0 (12 words), yo at 16 (32 words), ye at 48 (32 words)
constexpr int SLOT_WORDS = 80;
constexpr int A_OFF = 0, YO_OFF = 16, YE_OFF = 48;
// slots: 0-3 / 4-7 the two values a row may publish per phase, 8 the base point's x (stays for the whole ladder),
// 9 the constant 1, 10 a dump for the six idle lanes of a row, 11 where a whole element is laid down for my_limb,
// 12 the constant 1 / (2d) of the fixed-base walk's starting point, 13 the constant 2d (window-table rows)
constexpr int SLOT_X1 = 8, SLOT_ONE = 9, SLOT_DUMP = 10, SLOT_TMP = 11, SLOT_KDI = 12, SLOT_K2D = 13, NSLOTS = 14;
// ... and, behind the slots, the 32 table-row limbs a lane has fetched for the fixed-base walk (word j of lane l at j * 64 + l)
constexpr int ROWQ_OFF = NSLOTS * SLOT_WORDS, LDS_WORDS = ROWQ_OFF + 32 * 64;

struct Lane {
    u32 c, row;                    // column within the row (0..15; 0..9 hold limbs), row (0..3)
    u32 w, mask, mask_next;        // bits of limb c, its mask, the mask of limb c + 1
    u32 sh;                        // c & 1: odd limbs are doubled in the ye form
    u32 m1, m2;                    // 19 in lane 0 / in lanes 0 and 1, else 0: the carry's wrap-around
    u32 p2;                        // limb c of 2p (biased subtraction)
    u32 wr;                        // LDS word offset a lane adds to what it stores: 0, or the way to the dump slot
    bool odd_row, upper;           // row & 1, row >= 2
};

C25519_DEV Lane make_lane(u32 lane)
{
    Lane L;
    L.c = lane & 15;
    L.row = lane >> 4;
    const bool active = L.c < 10;
    L.sh = L.c & 1;
    L.w = L.sh ? 25 : 26;
    L.mask = L.sh ? M25 : M26;
    L.mask_next = L.sh ? M26 : M25;
    L.m1 = L.c == 0 ? 19u : 0u;
    L.m2 = L.c < 2 ? 19u : 0u;
    L.p2 = L.c == 0 ? 0x7ffffdau : (L.sh ? 0x3fffffeu : 0x7fffffeu);
    L.wr = active ? 0u : 0x80000000u;                     // flag: stores of idle lanes go to the dump slot
    L.odd_row = (L.row & 1) != 0;
    L.upper = L.row >= 2;
    return L;
}

// DPP moves within a 16-lane row (zero where nothing arrives)
C25519_DEV u32 row_shr1(u32 x) { return (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, true); }
C25519_DEV u32 row_shr2(u32 x) { return (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, true); }
C25519_DEV u32 row_ror7(u32 x) { return (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x127, 0xf, 0xf, true); }   // lane 0 <- lane 9
C25519_DEV u32 row_ror8(u32 x) { return (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x128, 0xf, 0xf, true); }   // lanes 0, 1 <- lanes 8, 9

// the value of the even row of each row pair in `even`, of the odd row in `odd`, in both rows of the pair
C25519_DEV void pair_exchange(u32& even, u32& odd, u32 v)
{
    const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
    even = r[0];
    odd = r[1];
}
// the value of rows 0, 1 in `lower`, of rows 2, 3 in `upper` (row r and row r + 2 see the pair (row r mod 2)'s values)
C25519_DEV void half_exchange(u32& lower, u32& upper, u32 v)
{
    const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    lower = r[0];
    upper = r[1];
}

// nothing moves across this point (the compiler sees one thread; the LDS traffic below is between lanes)
C25519_DEV void wave_fence()
{
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
}

// column sum -> reduced limb
// ... as ONE asm statement (valu_gfx950.cuh: row_carry -- one hazard slot instead of the compiler's three).  Behind a product
// level the compiler's own schedule is as good or better (it fills the slots with the next level's address arithmetic: +0.4 ..
// +0.8 us per single call with the asm there); where a carry follows a carry -- the base-point ladder's "times 9" -- the asm
// takes 7.7 us off one curve25519_dh_CalculatePublicKey (A/B on one box: tools/scratch/ab_single.py, profiles/r06_single_call_floor.txt).
C25519_DEV u32 carry_packed(const Lane& L, u64 S)
{
    return row_carry(S, L.w, L.mask, L.mask_next, L.m1, L.m2);
}
C25519_DEV u32 carry(const Lane& L, u64 S)
{
    const u32 l0 = (u32)S & L.mask;
    const u32 l1 = (u32)(S >> L.w) & L.mask_next;
    const u32 l2 = (u32)(S >> 51);
    u32 limb = l0 + row_shr1(l1) + row_shr2(l2) + row_ror7(l1) * L.m1 + row_ror8(l2) * L.m2;
    const u32 e = limb >> L.w;
    limb = (limb & L.mask) + row_shr1(e) + row_ror7(e) * L.m1;
    return limb;
}

// the same for a small sum (S < 2^46: a limb-wise sum, difference or a24 step): one piece moves up, limbs < 2^w + 2^25
C25519_DEV u32 carry_small(const Lane& L, u64 S)
{
    const u32 l1 = (u32)(S >> L.w);
    return ((u32)S & L.mask) + row_shr1(l1) + row_ror7(l1) * L.m1;
}

// zero `words` words of the kernel's LDS (operand forms of secret intermediates, fetched table rows) before it returns
C25519_DEV void wipe(u32* lds, int words /* a multiple of 4; lds 16-byte aligned */)
{
    wave_fence();
    uint4* p = reinterpret_cast<uint4*>(lds);
    for (int i = (int)threadIdx.x; i < words / 4; i += 64) p[i] = make_uint4(0, 0, 0, 0);
    wave_fence();
}
// ... behind the completion word, if the caller waits for one (thread 0 has stored the results)
C25519_DEV void finish(u32* lds, int words, const DoneWord* done)
{
    if (done && threadIdx.x == 0) signal_done(*done);
    wipe(lds, words);
}

// ---- SHA-512 on TWO waves -------------------------------------------------------------------------------------------------
// The hashes of a signature are three compressions in a row on the call's critical path, each 80 rounds of 47 instructions of
// which 19 are the message schedule -- work that does not depend on the round state.  A second wave of the workgroup computes
// it: wave 0 publishes the block's sixteen words, runs rounds 0..15 on its own copy while the helper expands words 16..31 (K
// added), and from there every sixteen rounds take their words ready-made from LDS (sha512_rounds16_wk: 28 instructions a round)
// while the helper is a chunk ahead: five workgroup barriers a block, ~2 300 instructions on wave 0 instead of ~3 750.  The
// helper must serve exactly the compressions wave 0 performs, in order (sha512_blocks counts them); between hashes it waits at a
// barrier and issues nothing.  wk: 80 64-bit words of LDS.
struct ShaTwoWaves {
    u64* wk;
    C25519_DEV void compress(u64 (&st)[8], u64 (&w)[16]) const
    {
        if ((threadIdx.x & 63u) == 0) {
#pragma unroll
            for (int i = 0; i < 16; i++) wk[i] = w[i];
        }
        __syncthreads();                                  // the helper may read the block
        u64 v[8];
#pragma unroll
        for (int i = 0; i < 8; i++) v[i] = st[i];
        sha512_rounds16<false>(v, w, 0);
#pragma unroll 1
        for (int r = 16; r < 80; r += 16) {
            __syncthreads();                              // words r .. r + 15 are there
            sha512_rounds16_wk(v, wk + r);
        }
#pragma unroll
        for (int i = 0; i < 8; i++) st[i] += v[i];
    }
};
// the helper wave: `blocks` compressions' schedules, then it is done
C25519_DEV void sha_schedule_server(u64* wk, int blocks)
{
#pragma unroll 1
    for (int b = 0; b < blocks; b++) {
        __syncthreads();
        u64 w[16];
#pragma unroll
        for (int i = 0; i < 16; i++) w[i] = wk[i];
#pragma unroll 1
        for (int r = 16; r < 80; r += 16) {
            sha512_schedule16_wk(w, wk + r, r);
            __syncthreads();
        }
    }
}

// a value's word offset in LDS for this lane's stores (idle lanes: the dump slot)
C25519_DEV u32 store_base(const Lane& L, u32 slot) { return ((L.wr ? (u32)SLOT_DUMP : slot)) * SLOT_WORDS; }

C25519_DEV void put_a(u32* lds, const Lane& L, u32 slot, u32 v) { lds[store_base(L, slot) + A_OFF + L.c] = v; }
C25519_DEV void put_y(u32* lds, const Lane& L, u32 slot, u32 v)
{
    u32* s = lds + store_base(L, slot);
    const u32 v19 = v * 19u, d = v << L.sh, d19 = v19 << L.sh;
    s[YO_OFF + L.c] = v19;
    s[YO_OFF + L.c + 10] = v;
    s[YE_OFF + L.c] = d19;
    s[YE_OFF + L.c + 10] = d;
}
C25519_DEV void put(u32* lds, const Lane& L, u32 slot, u32 v)
{
    put_a(lds, L, slot, v);
    put_y(lds, L, slot, v);
}

// column L.c of (value in slot xs) * (value in slot ys)
C25519_DEV u64 column(const u32* lds, const Lane& L, u32 xs, u32 ys)
{
    const u32* a = lds + xs * SLOT_WORDS + A_OFF;
    const u32* yo = lds + ys * SLOT_WORDS + YO_OFF + L.c + 1;
    const u32* ye = lds + ys * SLOT_WORDS + YE_OFF + L.c + 1;
    u64 acc = 0;
#pragma unroll
    for (int t = 0; t < 10; t++) acc += (u64)a[9 - t] * ((t & 1) ? yo[t] : ye[t]);
    return acc;
}

// one product level: every row multiplies the values of the two slots it names
C25519_DEV u32 mul_level(const u32* lds, const Lane& L, u32 xs, u32 ys)
{
    wave_fence();
    const u64 S = column(lds, L, xs, ys);
    wave_fence();
    return carry(L, S);
}

// per-row choice of a small constant: rows 0..3 take k0..k3
C25519_DEV u32 by_row(const Lane& L, u32 k0, u32 k1, u32 k2, u32 k3)
{
    return L.upper ? (L.odd_row ? k3 : k2) : (L.odd_row ? k1 : k0);
}

// limb L.c of a field element every lane holds whole: through LDS (lane 0 lays the ten limbs down, every lane picks
// its own), because a chain of ten selects on the lane's column is turned into a scratch array by the compiler
C25519_DEV u32 my_limb(u32* lds, const Lane& L, const fe& f)
{
    wave_fence();
    if (L.c == 0 && L.row == 0) {
#pragma unroll
        for (int i = 0; i < 10; i++) lds[SLOT_TMP * SLOT_WORDS + i] = f.v[i];
    }
    wave_fence();
    const u32 v = lds[SLOT_TMP * SLOT_WORDS + (L.c < 10 ? L.c : 0)];
    wave_fence();
    return v;
}

// every lane reads the ten limbs slot `s` holds in A-form
C25519_DEV void get_fe(fe& f, const u32* lds, u32 s)
{
#pragma unroll
    for (int i = 0; i < 10; i++) f.v[i] = lds[s * SLOT_WORDS + A_OFF + i];
}

// v <- v^2 in every row (each row on its own copy: slot = row), n times
C25519_DEV u32 sqr_n(u32* lds, const Lane& L, u32 v, int n)
{
#pragma unroll 1
    for (int i = 0; i < n; i++) {
        put(lds, L, L.row, v);
        v = mul_level(lds, L, L.row, L.row);
    }
    return v;
}
// v * w in every row (slots row and 4 + row)
C25519_DEV u32 mul2(u32* lds, const Lane& L, u32 v, u32 w)
{
    put_a(lds, L, L.row, v);
    put_y(lds, L, 4 + L.row, w);
    return mul_level(lds, L, L.row, 4 + L.row);
}

// z^(2^250 - 1) and z^11 limb-per-lane: the shared head of the two exponentiations (fe_chain250, fe25519.cuh)
C25519_DEV u32 chain250(u32* lds, const Lane& L, u32 z, u32& x11)
{
    const u32 x2 = sqr_n(lds, L, z, 1);
    u32 t = sqr_n(lds, L, x2, 2);
    const u32 x9 = mul2(lds, L, t, z);
    x11 = mul2(lds, L, x9, x2);
    t = sqr_n(lds, L, x11, 1);
    const u32 x5 = mul2(lds, L, t, x9);                   // z^(2^5 - 1)
    t = sqr_n(lds, L, x5, 5);
    const u32 x10 = mul2(lds, L, t, x5);
    t = sqr_n(lds, L, x10, 10);
    const u32 x20 = mul2(lds, L, t, x10);
    t = sqr_n(lds, L, x20, 20);
    t = mul2(lds, L, t, x20);
    t = sqr_n(lds, L, t, 10);
    const u32 x50 = mul2(lds, L, t, x10);
    t = sqr_n(lds, L, x50, 50);
    const u32 x100 = mul2(lds, L, t, x50);
    t = sqr_n(lds, L, x100, 100);
    t = mul2(lds, L, t, x100);
    t = sqr_n(lds, L, t, 50);
    return mul2(lds, L, t, x50);                          // z^(2^250 - 1)
}

// z^(p-2) limb-per-lane: the 254 S + 11 M chain of fe_invert_fermat (ecp_Inverse, curve25519_mehdi.c:340-409); 0 -> 0
C25519_DEV u32 invert_fermat(u32* lds, const Lane& L, u32 z)
{
    u32 x11;
    u32 t = chain250(lds, L, z, x11);
    t = sqr_n(lds, L, t, 5);
    return mul2(lds, L, t, x11);
}

// 1 / z of every row's value (0 -> 0), limb per lane in and out.  The chain above is 265 product levels of ~380 cycles for a lone
// wave (44 us); the division steps of safegcd25519.cuh have no use for sixty-four lanes but are ~14 800 instructions on ONE
// (~33 us): every lane of a row picks up the row's whole element, runs them on it (the sixteen lanes of a row on the same value,
// the four rows each on their own), and takes its limb of the result back through the row's slot.
C25519_DEV u32 invert(u32* lds, const Lane& L, u32 z)
{
#if C25519_INVERT_SAFEGCD
    put_a(lds, L, L.row, z);
    wave_fence();
    fe f, r;
    get_fe(f, lds, L.row);
    wave_fence();
    fe_invert_quad(r, f);                                  // (the row's sixteen lanes hold f: every quad of them does)
    if (L.c == 0) {
#pragma unroll
        for (int i = 0; i < 10; i++) lds[L.row * SLOT_WORDS + A_OFF + i] = r.v[i];
    }
    wave_fence();
    const u32 v = lds[L.row * SLOT_WORDS + A_OFF + (L.c < 10 ? L.c : 0)];
    wave_fence();
    return v;
#else
    return invert_fermat(lds, L, z);
#endif
}

// z^((p-5)/8) = z^(2^252 - 3)   (fe_pow2523)
C25519_DEV u32 pow2523(u32* lds, const Lane& L, u32 z)
{
    u32 x11;
    u32 t = chain250(lds, L, z, x11);
    t = sqr_n(lds, L, t, 2);
    return mul2(lds, L, t, z);
}

// x from y with the requested parity, by the whole wave: ge_calc_x_checked (ge25519.cuh; ed25519_CalculateX) limb per lane, all four
// rows on the same values -- 262 product levels in a row instead of one lane's 27 000 instructions.  yl: this lane's limb of y (the
// same y in every row); xl: the limb of x; x_zero: all-ones iff x == 0 (the one x whose parity cannot be chosen).  Returns
// all-ones iff v x^2 == u, i.e. the point is on the curve -- like the per-lane code there is no rejection: a non-square just
// yields what the formula yields.  Uses SLOT_X1 and SLOT_K2D for the constants d and sqrt(-1) (the walks set theirs afterwards).
C25519_DEV u32 calc_x_checked(u32* lds, const Lane& L, u32& xl, u32& x_zero, u32 yl, u32 parity)
{
    const u32 one = L.c == 0 ? 1u : 0u;
    put_y(lds, L, SLOT_X1, my_limb(lds, L, fe_const(K_D)));
    put_y(lds, L, SLOT_K2D, my_limb(lds, L, fe_const(K_SQRTM1)));
    auto times = [&](u32 v, u32 slot) -> u32 { put_a(lds, L, L.row, v); return mul_level(lds, L, L.row, slot); };
    const u32 y2 = sqr_n(lds, L, yl, 1);
    const u32 v = times(y2, SLOT_X1) + one;               // d y^2 + 1
    const u32 u = carry_small(L, (u64)(y2 + L.p2 - one)); // y^2 - 1, reduced
    u32 b = sqr_n(lds, L, v, 1);
    u32 a = mul2(lds, L, u, b);
    a = mul2(lds, L, a, v);                               // u v^3
    b = sqr_n(lds, L, b, 1);                              // v^4
    b = mul2(lds, L, a, b);                               // u v^7
    b = pow2523(lds, L, b);
    u32 x = mul2(lds, L, b, a);
    // is v x^2 == u (x is the root) or == -u (x sqrt(-1) is)?  Every lane reads both values whole and reduces them itself.
    u32 c = sqr_n(lds, L, x, 1);
    c = mul2(lds, L, c, v);
    put_a(lds, L, L.row, c + u);
    put_a(lds, L, 4 + L.row, c + L.p2 - u);
    wave_fence();
    fe S, D;
    get_fe(S, lds, 0);
    get_fe(D, lds, 4);
    u32 sw[8], dw[8];
    fe_to_words(sw, S);
    fe_to_words(dw, D);
    u32 nz = 0, nz2 = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) { nz |= dw[i]; nz2 |= sw[i]; }
    wave_fence();                                         // (every lane has read the two sums: the slots may be written again)
    const u32 xi = times(x, SLOT_K2D);
    x = nz ? xi : x;                                      // :92-93
    put_a(lds, L, L.row, x);
    wave_fence();
    fe X;
    get_fe(X, lds, 0);
    u32 xw[8];
    fe_to_words(xw, X);                                   // canonical, to read the parity (:95-99)
    u32 any = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) any |= xw[i];
    const u32 neg = carry_small(L, (u64)(L.p2 - x));
    xl = ((xw[0] ^ parity) & 1u) ? neg : x;
    x_zero = any ? 0u : 0xffffffffu;
    wave_fence();
    return (nz == 0 || nz2 == 0) ? 0xffffffffu : 0u;
}

// (X : Z) <- 2 (X : Z) for a point held x in the even rows, z in the odd rows (both row pairs may hold one): two product
// levels, {(x+z)^2, (x-z)^2} and {x' = AA * BB, z' = E * (AA + 121665 E)}   (ecp_MontDouble, curve25519_dh.c:40-54)
C25519_DEV u32 mont_double(u32* lds, const Lane& L, u32 v)
{
    u32 ev, od;
    pair_exchange(ev, od, v);
    const u32 val = L.odd_row ? ev + L.p2 - od : ev + od;
    put(lds, L, L.row, val);
    v = mul_level(lds, L, L.row, L.row);                   // even rows AA, odd rows BB
    pair_exchange(ev, od, v);
    const u32 E = ev + L.p2 - od;
    const u32 F = carry_small(L, (u64)E * 121665u + ev);
    put_a(lds, L, L.row, L.odd_row ? E : ev);
    put_y(lds, L, L.row, L.odd_row ? F : od);
    return mul_level(lds, L, L.row, L.row);
}

// One ladder step on the state (row 0: x of the sum, row 1: its z, row 2: x of the double, row 3: its z), limb per lane.
// eq: all-ones when this bit equals the previous one (the doubling then continues from the double, else from the sum).
template <bool BASE9>
C25519_DEV u32 ladder_step(u32* lds, const Lane& L, u32 v, u32 eq)
{
    // phase 0: B = x1+z1 (row 0), A = x1-z1 (row 1), Dp = x2+z2 (row 2), C = x2-z2 (row 3); the doubling's inputs
    u32 ev, od;
    pair_exchange(ev, od, v);
    const u32 sum = ev + od, diff = ev + L.p2 - od;
    const u32 val = L.odd_row ? diff : sum;
    u32 lo, hi;
    half_exchange(lo, hi, val);                            // rows 2, 3 see (B, A) in lo and their own (Dp, C) in hi
    const u32 sel = lo ^ ((lo ^ hi) & eq);                 // P = x+z (row 2) and M = x-z (row 3) of the point to double
    put(lds, L, L.row, val);
    put(lds, L, 4 + L.row, sel);
    // level 1: row 0 A*Dp, row 1 C*B, row 2 P^2, row 3 M^2
    v = mul_level(lds, L, by_row(L, 1, 3, 6, 7), by_row(L, 2, 0, 6, 7));
    // phase 1.5: DA+CB (row 0), DA-CB (row 1), F = AA + 121665 E (row 2), E = AA-BB (row 3) -- all through one small carry
    pair_exchange(ev, od, v);
    const u32 d2 = ev + L.p2 - od;
    const u32 base = L.upper ? (L.odd_row ? d2 : ev) : (L.odd_row ? d2 : ev + od);
    const u32 k = (L.upper && !L.odd_row) ? 121665u : 0u;
    const u32 w = carry_small(L, (u64)d2 * k + base);
    put(lds, L, L.row, v);                                 // slots 2, 3: AA, BB
    put(lds, L, 4 + L.row, w);                             // slots 4..7: DA+CB, DA-CB, F, E
    // level 2: row 0 (DA+CB)^2 = x3, row 1 (DA-CB)^2, row 2 AA*BB = x4, row 3 E*F = z4
    v = mul_level(lds, L, by_row(L, 4, 5, 2, 7), by_row(L, 4, 5, 3, 6));
    // level 3: row 1 times the base point's x (times 9: a small constant), the other rows times one
    put_a(lds, L, L.row, v);
    if (BASE9) {
        wave_fence();
        const u32 k9 = (!L.upper && L.odd_row) ? 9u : 1u;
        return carry_packed(L, (u64)v * k9);
    }
    return mul_level(lds, L, L.row, by_row(L, SLOT_ONE, SLOT_X1, SLOT_ONE, SLOT_ONE));
}

// ---- the ladder on TWO waves: a step in two product levels instead of three --------------------------------------------------
// The third level of ladder_step is ONE product, z3 = x1 (DA - CB)^2, with three rows idle: the multiplication by the base point's x
// can only follow the squaring.  Carry the sum point TWICE -- S = (x : z) and x1 S = (x1 x : x1 z) -- and it disappears:
//   level 1   DA = A Dp, CB = C B, x1 DA = (x1 A) Dp, x1 CB = C (x1 B)      |  AA = P^2, BB = M^2
//   level 2   x3 = V^2, z3 = U (x1 U), x1 x3 = V (x1 V), x1 z3 = (x1 U)^2   |  x4 = AA BB, z4 = E F
// with V = DA + CB, U = DA - CB and x1 V, x1 U the same sums of the scaled products: x1 U^2 = U (x1 U), x1 (x1 U^2) = (x1 U)^2.
// Twelve products instead of nine -- the eight on the left are the differential addition, a full wave's two levels (state: x, z,
// x1 x, x1 z in the four rows); the four on the right are the doubling, which a SECOND wave carries (x in its even rows, z in
// the odd; its two row pairs square the double's and the sum's x + z, x - z side by side and the scalar's bit picks one).  Within a step neither wave needs the other; at the step's start wave 0 publishes B = x + z and A = x - z of the
// sum (the doubling continues from them when the scalar's bit flips) and wave 1 publishes Dp and C of the double: ONE workgroup
// barrier per step.  The published slots alternate between two sets by the step's parity, so a wave that is a step ahead writes
// where nobody reads.  (Slot 10 stays the dump of the idle lanes.)
constexpr int X2_PUB = 12;                                 // + 6 * parity: B, A, x1 B, x1 A (wave 0), Dp, C (wave 1)
constexpr int X2_SHARED_SLOTS = 24;
constexpr int X2_LDS_WORDS = (X2_SHARED_SLOTS + 2 * NSLOTS) * SLOT_WORDS;    // ... and a private region per wave

// wave 0: the sum point and x1 times it (rows: x, z, x1 x, x1 z) through one step.  sh: the shared slots, lds: this wave's own.
C25519_DEV u32 ladder2_step_sum(u32* sh, u32* lds, const Lane& L, u32 v, u32 parity)
{
    const u32 pub = X2_PUB + 6 * parity;
    u32 ev, od;
    pair_exchange(ev, od, v);
    const u32 val = L.odd_row ? ev + L.p2 - od : ev + od;  // B, A, x1 B, x1 A
    put(sh, L, pub + L.row, val);
    __syncthreads();
    // level 1: A * Dp, C * B, (x1 A) * Dp, C * (x1 B)
    v = mul_level(sh, L, L.odd_row ? pub + 5 : pub + 1 + (L.upper ? 2 : 0), L.odd_row ? pub + (L.upper ? 2 : 0) : pub + 4);
    pair_exchange(ev, od, v);
    const u32 w = carry_small(L, (u64)(L.odd_row ? ev + L.p2 - od : ev + od));   // V, U, x1 V, x1 U
    put(lds, L, L.row, w);
    // level 2: V^2 = x3, U * (x1 U) = z3, V * (x1 V) = x1 x3, (x1 U)^2 = x1 z3
    return mul_level(lds, L, by_row(L, 0, 1, 0, 3), by_row(L, 0, 3, 2, 3));
}

// wave 1: the double (even rows x, odd rows z; both row pairs alike) through the same step
C25519_DEV u32 ladder2_step_double(u32* sh, u32* lds, const Lane& L, u32 v, u32 eq, u32 parity)
{
    const u32 pub = X2_PUB + 6 * parity;
    u32 ev, od;
    pair_exchange(ev, od, v);
    const u32 val = L.odd_row ? ev + L.p2 - od : ev + od;  // Dp, C
    put(sh, L, pub + 4 + (L.odd_row ? 1 : 0), val);
    __syncthreads();
    // the point to double is this one when the bit repeats, else the sum: the lower row pair squares Dp and C, the upper pair B and
    // A as wave 0 published them (fixed slots: no secret-dependent address), and the bit chooses between the two pairs' squares
    const u32 s = (L.upper ? pub : pub + 4) + (L.odd_row ? 1 : 0);
    v = mul_level(sh, L, s, s);
    u32 lo, hi;
    half_exchange(lo, hi, v);
    v = hi ^ ((hi ^ lo) & eq);                             // AA = P^2 (even rows), BB = M^2 (odd rows), in both row pairs
    pair_exchange(ev, od, v);
    const u32 E = ev + L.p2 - od;
    const u32 F = carry_small(L, (u64)E * 121665u + ev);
    put_a(lds, L, L.row, L.odd_row ? E : ev);
    put_y(lds, L, L.row, L.odd_row ? F : od);
    return mul_level(lds, L, L.row, L.row);                // x4 = AA * BB, z4 = E * F
}

// ---- the fixed-base Edwards walk, one operation per wave ------------------------------------------------------------
// A point (X : Y : Z : T) lives in the four rows (row 0 X ... row 3 T), limb per lane; an addition of a precomputed affine
// row (edp_AddAffinePoint, ed25519_sign.c:97-115) and a doubling (edp_DoublePoint, :122-143) are two product levels each:
// {(Y-X) ymx, (Y+X) ypx, 2Z, T t2d} then {F E, G H, F G, E H}, and {X^2, Y^2, (X+Y)^2, Z^2} then {E Fn, G Hn, G Fn, E Hn}
// with ge25519.cuh's sign conventions (Hn = A + B, Fn = 2 Z^2 + A - B: all four outputs negated, the same point).
// The walk is ge_base_mult's: eight signed comb tables, 3 doublings, 31 additions -- the same point as the reference's
// 31-doubling walk (edp_BasePointMult, ed25519_sign.c:215-244), hence the same bytes after the inversion.

// limb L.c of field `f` (0 ypx, 1 ymx, 2 t2d) of the row a column byte selects in one signed comb table (limb-major
// [30][128] in device memory); a negative column swaps ypx / ymx and negates t2d (ge25519.cuh: lds_load_pa_signed)
C25519_DEV u32 row_limb(const u32* __restrict__ tbl, const Lane& L, u32 colbyte, u32 f)
{
    const u32 neg = ((colbyte >> 7) & 1u) - 1u;           // all-ones: negative column
    const u32 row = (colbyte ^ neg) & 127u;
    const u32 ff = (f < 2 && neg) ? 1u - f : f;
    const u32 c = L.c < 10 ? L.c : 9;
    const u32 wd = tbl[(ff * 10 + c) * BASE_ROWS + row];
    return (f == 2 && neg) ? L.p2 - wd : wd;
}
// ... of the field an addition's first level multiplies THIS row by: row 0 ymx, row 1 ypx, row 3 t2d (row 2: unused)
C25519_DEV u32 row_limb_for_add(const u32* __restrict__ tbl, const Lane& L, u32 colbyte)
{
    return row_limb(tbl, L, colbyte, by_row(L, 1, 0, 2, 2));
}

// second half of every addition: from (A, B, D, C) in the rows to the sum
C25519_DEV u32 ge_add_finish(u32* lds, const Lane& L, u32 v)
{
    u32 ev, od;
    pair_exchange(ev, od, v);
    const u32 w = L.odd_row ? ev + od : (L.upper ? ev + L.p2 - od : od + L.p2 - ev);   // E = B-A, H = B+A, F = D-C, G = D+C
    put(lds, L, L.row, w);
    return mul_level(lds, L, by_row(L, 2, 3, 2, 0), by_row(L, 0, 1, 3, 1));          // F E, G H, F G, E H
}

// p += q, q a table row given as the limb each lane multiplies by (row_limb_for_add)
C25519_DEV u32 ge_add(u32* lds, const Lane& L, u32 v, u32 qlimb)
{
    u32 ev, od;
    pair_exchange(ev, od, v);                             // lower pair: X, Y; upper pair: Z, T
    const u32 op = L.upper ? (L.odd_row ? od : ev + ev) : (L.odd_row ? ev + od : od + L.p2 - ev);
    put_a(lds, L, L.row, op);                             // Y-X, Y+X, 2Z, T
    put_y(lds, L, 4 + L.row, qlimb);
    v = mul_level(lds, L, L.row, by_row(L, 4, 5, SLOT_ONE, 7));      // A, B, D, C
    return ge_add_finish(lds, L, v);
}

// p += q (or p -= q), q a projective precomputed row (Y+X, Y-X, 2dT, 2Z: edp_AddPoint, ed25519_verify.c:142-161) whose four
// fields already sit in LDS in multiplier form, slots s_ypx .. s_ypx + 3.  -q swaps the first two and negates 2dT, which
// costs nothing here: the two rows read each other's slot and row 3 multiplies -T instead.
C25519_DEV u32 ge_add_pe(u32* lds, const Lane& L, u32 v, u32 s_ypx, u32 neg)
{
    u32 ev, od;
    pair_exchange(ev, od, v);
    const u32 t = neg ? L.p2 - od : od;
    const u32 op = L.upper ? (L.odd_row ? t : ev) : (L.odd_row ? ev + od : od + L.p2 - ev);
    put_a(lds, L, L.row, op);                             // Y-X, Y+X, Z, +-T
    const u32 sw = neg ? 1u : 0u;
    v = mul_level(lds, L, L.row, s_ypx + by_row(L, 1 ^ sw, 0 ^ sw, 3, 2));           // A, B, D, C
    return ge_add_finish(lds, L, v);
}

// p = 2p
C25519_DEV u32 ge_dbl(u32* lds, const Lane& L, u32 v)
{
    u32 ev, od, lo, hi;
    pair_exchange(ev, od, v);
    half_exchange(lo, hi, ev + od);                       // lo: X + Y, in every row
    put(lds, L, L.row, L.upper ? (L.odd_row ? ev : lo) : v);          // X, Y, X+Y, Z
    v = mul_level(lds, L, L.row, L.row);                  // A, B, (X+Y)^2, Z^2
    pair_exchange(ev, od, v);
    const u32 hg = L.odd_row ? od + L.p2 - ev : ev + od;  // rows 0, 1: Hn = A + B, G = B - A
    half_exchange(lo, hi, hg);                            // lo: Hn in the even rows, G in the odd rows
    const u32 k = L.odd_row ? 2u : 1u;
    const u32 w = carry_small(L, L.upper ? v * k + 2 * L.p2 - lo : hg);   // Hn, G, E = (X+Y)^2 - Hn, Fn = 2 Z^2 - G
    put(lds, L, L.row, w);
    return mul_level(lds, L, by_row(L, 2, 1, 1, 2), by_row(L, 3, 0, 3, 0));           // E Fn, G Hn, G Fn, E Hn
}

// S = k * B for a scalar below 2^255 + 2^254 (clamped, or reduced mod L), tbl = the eight signed comb tables in device
// memory.  SLOT_ONE must hold the constant one.  Every row fetch is issued before the walk starts (the addresses depend on
// the scalar alone), so the walk itself never waits for memory.
C25519_DEV u32 ge_base_mult(u32* lds, const Lane& L, const u32 (&k)[8], const u32* __restrict__ tbl)
{
    u32 w[8];
    sc_signed_comb(w, k);
    const u32 lane = L.row * 16 + L.c;
#pragma unroll
    for (int m = 0; m < BASE_STEP; m++)
#pragma unroll
        for (int t = 0; t < BASE_NT; t++)
            lds[ROWQ_OFF + (m * BASE_NT + t) * 64 + lane] = row_limb_for_add(tbl + t * BASE_TBL_WORDS, L, fold8_at(w, t * BASE_STEP + m));
    // the start: (2x, 2y, 2, 2xy) of table 0's row (ed25519_sign.c:226-230 with R = 1): ypx - ymx, ypx + ymx, 2, t2d / (2d)
    const u32 c0 = fold8_at(w, 0);
    const u32 ypx = row_limb(tbl, L, c0, 0), ymx = row_limb(tbl, L, c0, 1), t2d = row_limb(tbl, L, c0, 2);
    put_y(lds, L, SLOT_KDI, my_limb(lds, L, fe_const(K_DI)));
    const u32 two = L.c == 0 ? 2u : 0u;
    put_a(lds, L, L.row, L.upper ? (L.odd_row ? t2d : two) : (L.odd_row ? ypx + ymx : ypx + L.p2 - ymx));
    u32 v = mul_level(lds, L, L.row, by_row(L, SLOT_ONE, SLOT_ONE, SLOT_ONE, SLOT_KDI));
#pragma unroll 1
    for (int m = 0; m < BASE_STEP; m++) {
        if (m) v = ge_dbl(lds, L, v);
#pragma unroll 1
        for (int t = m ? 0 : 1; t < BASE_NT; t++) {
            v = ge_add(lds, L, v, lds[ROWQ_OFF + (m * BASE_NT + t) * 64 + lane]);
        }
    }
    return v;
}

// The same walk over the WIDE comb (ge25519.cuh: 13 teeth 20 bits apart, four packed 4096-row tables): 19 additions + 4
// doublings = 46 product levels instead of 68.  A packed row is Y+X | Y-X | 2dT as 255-bit integers; a lane takes the limb
// of the field its row multiplies by straight out of the row's words (packed_limb).
C25519_DEV u32 packed_limb(const u32* __restrict__ p, const Lane& L);
C25519_DEV u32 wide_row_limb(const u32* __restrict__ tbl, const Lane& L, u32 col, u32 f)
{
    const u32 neg = ((col >> (WB_TEETH - 1)) & 1u) - 1u;  // all-ones: negative column
    const u32 row = (col ^ neg) & (u32)(WB_ROWS - 1);
    const u32 ff = (f < 2 && neg) ? 1u - f : f;
    const u32 wd = packed_limb(tbl + (size_t)row * WB_ROW_WORDS + 8 * ff, L);
    return (f == 2 && neg) ? L.p2 - wd : wd;
}

// blind (or null): a blinding context's 48 words (lanes.cuh: bl, zr, BP) -- the walk then runs on (k + bl) mod L from a starting
// point spread over its projective class by zr, and BP comes on top (edp_BasePointMultiply with a context, ed25519_sign.c:254-259)
C25519_DEV u32 ge_base_mult_wide(u32* lds, const Lane& L, const u32 (&k_in)[8], const u32* __restrict__ wide,
                                 const u32* __restrict__ blind = nullptr)
{
    u32 k[8];
#pragma unroll
    for (int i = 0; i < 8; i++) k[i] = k_in[i];
    if (blind) {
        u32 bl[8];
#pragma unroll
        for (int i = 0; i < 8; i++) bl[i] = blind[i];
        sc_add(k, k_in, bl);                              // 256 bits, congruent to k + bl mod L (eco_AddReduce :255)
        sc_mod(k);
    }
    u32 cols[WB_COLS];                                    // (compile-time indices below: registers)
    wb_columns(cols, 1, k);
    // every row fetch is issued before the walk starts (the addresses depend on the scalar alone), the first row's first: the walk
    // below is unrolled, the rows stay in registers, and an addition waits for ITS row only -- the later rows arrive under the
    // earlier additions (fetched into LDS in front of a rolled loop, the first level waited ~2 us for all twenty)
    const u32 ypx = wide_row_limb(wide, L, cols[0], 0), ymx = wide_row_limb(wide, L, cols[0], 1), t2d = wide_row_limb(wide, L, cols[0], 2);
    u32 rowv[WB_COLS];
#pragma unroll
    for (int s = 1; s < WB_COLS; s++)
        rowv[s] = wide_row_limb(wide + (size_t)(s % WB_NT) * WB_ROWS * WB_ROW_WORDS, L, cols[s], by_row(L, 1, 0, 2, 2));
    put_y(lds, L, SLOT_KDI, my_limb(lds, L, fe_const(K_DI)));
    const u32 two = L.c == 0 ? 2u : 0u;
    put_a(lds, L, L.row, L.upper ? (L.odd_row ? t2d : two) : (L.odd_row ? ypx + ymx : ypx + L.p2 - ymx));
    u32 v = mul_level(lds, L, L.row, by_row(L, SLOT_ONE, SLOT_ONE, SLOT_ONE, SLOT_KDI));
    if (blind) {                                          // (2x, 2y, 2, 2xy) * zr: the same point, another representative
        put_a(lds, L, L.row, v);
        put_y(lds, L, 4, packed_limb(blind + 8, L));
        v = mul_level(lds, L, L.row, 4);
    }
#pragma unroll
    for (int m = 0; m < WB_STEP; m++) {
        if (m) v = ge_dbl(lds, L, v);
#pragma unroll
        for (int t = m ? 0 : 1; t < WB_NT; t++)
            v = ge_add(lds, L, v, rowv[m * WB_NT + t]);
    }
    if (blind) {                                          // + BP, a precomputed projective point (Y+X, Y-X, 2dT, 2Z)
#pragma unroll
        for (int f = 0; f < 4; f++) put_y(lds, L, 4 + f, packed_limb(blind + 16 + 8 * f, L));
        v = ge_add_pe(lds, L, v, 4, 0u);
    }
    return v;
}

// canonical (x, y) words of the point in the rows: one inversion of Z, two products, the batch kernels' encoding
C25519_DEV void ge_affine_words(u32 (&xw)[8], u32 (&yw)[8], u32* lds, const Lane& L, u32 v)
{
    u32 ev, od, lo, hi;
    pair_exchange(ev, od, v);
    half_exchange(lo, hi, ev);                            // hi: Z in every row
    const u32 zi = invert(lds, L, hi);
    const u32 r = mul2(lds, L, v, zi);                    // row 0 x, row 1 y
    put_a(lds, L, L.row, r);
    wave_fence();
    fe x, y;
    get_fe(x, lds, 0);
    get_fe(y, lds, 1);
    wave_fence();
    fe_to_words(xw, x);
    fe_to_words(yw, y);
}

// ---- verification: the lattice walk, one element per wave --------------------------------------------------------------
// W = sigma*B + tau*Q + rho*Rn (verify_fast.cuh: ge_walk_is_neutral) for ONE element: the two 9-row window tables the points
// kernel left in the element's scratch (packed rows), the biased scalars and sigma's comb columns the scalar kernel left
// there, the walk's signed comb table in device memory.  All 72 table fields are unpacked limb-per-lane into LDS
// multiplier forms first, all comb rows fetched, then the walk runs from LDS alone: two product levels per point
// operation instead of ~800 instructions of one lane.
constexpr int VSLOT0 = NSLOTS;                            // slot of field f of row r of table t: VSLOT0 + (t * 9 + r) * 4 + f
constexpr int V_ROWQ_OFF = (VSLOT0 + 2 * WTABLE_ROWS * 4) * SLOT_WORDS;
constexpr int V_LDS_WORDS = V_ROWQ_OFF + SC_ROUNDS * 4 * 64;

// limb L.c of the 255-bit integer at p[0..7] (fe_from_words: bit 255 counts 19)
C25519_DEV u32 packed_limb(const u32* __restrict__ p, const Lane& L)
{
    const u32 c = L.c < 10 ? L.c : 9;
    const u32 pos = 26 * c - (c >> 1);                     // 0, 26, 51, 77, 102, 128, 153, 179, 204, 230
    const u32 idx = pos >> 5;
    const u64 two = (u64)p[idx] | ((u64)p[idx < 7 ? idx + 1 : 7] << 32);
    u32 limb = (u32)(two >> (pos & 31)) & L.mask;
    limb += (L.c == 0) ? 19u * (p[7] >> 31) : 0u;
    return limb;
}

// limb of the field THIS row multiplies by in an addition of column byte c of the walk's signed comb (sign = bit
// SC_TEETH - 1 clear, ge_add_pa_comb): row 0 ymx, row 1 ypx, row 3 2dxy, swapped / negated for a negative column
C25519_DEV u32 comb_limb_for_add(const u32* __restrict__ tbl, const Lane& L, u32 c)
{
    const u32 neg = ((c >> (SC_TEETH - 1)) & 1u) - 1u;
    const u32 r = (c ^ neg) & (u32)(SC_ROWS - 1);
    const u32 f = by_row(L, 1, 0, 2, 2);
    const u32 ff = (f < 2 && neg) ? 1u - f : f;
    const u32 lc = L.c < 10 ? L.c : 9;
    const u32 wd = tbl[(ff * 10 + lc) * SC_ROWS + r];
    return (f == 2 && neg) ? L.p2 - wd : wd;
}

// The window table of ONE point -- rows 0 .. 8 times P in precomputed form (Y+X, Y-X, 2dT, 2Z), what wtable_build
// (verify_fast.cuh) packs into global memory one lane at a time -- built by the whole wave straight into the LDS multiplier
// forms the walk reads (slots VSLOT0 + (t * WTABLE_ROWS + r) * 4 + f): the same points by the same chain of doublings and "+ P"
// (2P, 3P, 4P = 2 (2P), 5P, 6P = 2 (3P), 7P, 8P = 2 (4P)), two product levels per operation and one per row for the
// conversion (2d T is a product; the other three fields ride the level times one, for its carry).  xl, yl: limb L.c of the
// point's affine x and y (the same in every row).  SLOT_ONE must hold one.
C25519_DEV void wtable_build_lds(u32* lds, const Lane& L, int t, u32 xl, u32 yl)
{
    const u32 one = L.c == 0 ? 1u : 0u;
    put_y(lds, L, SLOT_K2D, my_limb(lds, L, fe_const(K_2D)));
    const int base = VSLOT0 + t * WTABLE_ROWS * 4;
    auto store_row = [&](int r, u32 v) {
        u32 ev, od;
        pair_exchange(ev, od, v);                         // lower pair: X, Y; upper pair: Z, T
        const u32 val = L.upper ? (L.odd_row ? od : ev + ev) : (L.odd_row ? od + L.p2 - ev : ev + od);
        put_a(lds, L, L.row, val);                        // Y+X, Y-X, 2Z, T
        const u32 w = mul_level(lds, L, L.row, by_row(L, SLOT_ONE, SLOT_ONE, SLOT_ONE, SLOT_K2D));
        put_y(lds, L, base + r * 4 + by_row(L, 0, 1, 3, 2), w);          // fields: ypx, ymx, t2d, z2
    };
    // row 0, the neutral element: (1, 1, 0, 2)
    put_y(lds, L, base + by_row(L, 0, 1, 3, 2), L.upper ? (L.odd_row ? 0u : one + one) : one);
    // P = (x : y : 1 : x y)
    put_a(lds, L, L.row, L.upper ? (L.odd_row ? xl : one) : (L.odd_row ? yl : xl));
    put_y(lds, L, 4 + L.row, yl);
    const u32 v1 = mul_level(lds, L, L.row, by_row(L, SLOT_ONE, SLOT_ONE, SLOT_ONE, 7));
    store_row(1, v1);
    const u32 v2 = ge_dbl(lds, L, v1);
    store_row(2, v2);
    const u32 v3 = ge_add_pe(lds, L, v2, base + 4, 0u);
    store_row(3, v3);
    const u32 v4 = ge_dbl(lds, L, v2);
    store_row(4, v4);
    store_row(5, ge_add_pe(lds, L, v4, base + 4, 0u));
    const u32 v6 = ge_dbl(lds, L, v3);
    store_row(6, v6);
    store_row(7, ge_add_pe(lds, L, v6, base + 4, 0u));
    store_row(8, ge_dbl(lds, L, v4));
}

// ---- the three products of the lattice equation, one wave each (engine.hip: k_ed25519_verify_one_per_group) -------------
// sigma*B + tau*Q + rho*Rn = O is three independent scalar products; a workgroup of three waves computes them side by side
// -- each wave with an LDS region of its own (operand slots, its point's window table) -- and one wave adds them up.

// k * P for one biased 160-bit scalar (signed radix-16 digits, verify_fast.cuh) over P's window table in slots tbase ...
// (wtable_build_lds): the first digit's row as an extended point, then per digit four doublings and one row.
// word(w): word w of the scalar.  SLOT_ONE and SLOT_KDI must be set.
template <typename WordFn>
C25519_DEV u32 walk_point(u32* lds, const Lane& L, WordFn word, int tbase, int top)
{
    u32 v;
    {
        u32 neg;
        const u32 m = signed16_of(neg, word(top >> 3), top & 7);
        const u32* row = lds + (tbase + m * 4) * SLOT_WORDS + YO_OFF + 10 + (L.c < 10 ? L.c : 9);      // plain limbs of a field
        const u32 a = row[(neg ? 1 : 0) * SLOT_WORDS], b = row[(neg ? 0 : 1) * SLOT_WORDS];             // Y+X, Y-X of +-row
        const u32 t2d = row[2 * SLOT_WORDS], z2 = row[3 * SLOT_WORDS];
        wave_fence();
        put_a(lds, L, L.row, L.upper ? (L.odd_row ? (neg ? L.p2 - t2d : t2d) : z2) : (L.odd_row ? a + b : a + L.p2 - b));
        v = mul_level(lds, L, L.row, by_row(L, SLOT_ONE, SLOT_ONE, SLOT_ONE, SLOT_KDI));                 // 2x, 2y, 2z, 2xy
    }
#pragma unroll 1
    for (int i = top - 1; i >= 0; i--) {
#pragma unroll 1
        for (int j = 0; j < 4; j++) v = ge_dbl(lds, L, v);
        u32 neg;
        const u32 m = signed16_of(neg, word(i >> 3), i & 7);
        v = ge_add_pe(lds, L, v, tbase + m * 4, neg);
    }
    return v;
}

// limb L.c of field f (0 ypx, 1 ymx, 2 2dxy) of the row column c of the walk's signed comb selects (sign included)
C25519_DEV u32 comb_limb(const u32* __restrict__ tbl, const Lane& L, u32 c, u32 f)
{
    const u32 neg = ((c >> (SC_TEETH - 1)) & 1u) - 1u;
    const u32 r = (c ^ neg) & (u32)(SC_ROWS - 1);
    const u32 ff = (f < 2 && neg) ? 1u - f : f;
    const u32 lc = L.c < 10 ? L.c : 9;
    const u32 wd = tbl[(ff * 10 + lc) * SC_ROWS + r];
    return (f == 2 && neg) ? L.p2 - wd : wd;
}

// sigma * B over the walk's signed comb by Horner's rule: column SC_COLS - 1 first, then per column a doubling and a row
// (column c has weight 2^c).  sigma_word(w): the columns as sc_comb_columns packs them.  rowq: SC_ROUNDS * 4 * 64 words.
template <typename WordFn>
C25519_DEV u32 walk_comb(u32* lds, u32* rowq, const Lane& L, WordFn sigma_word, const u32* __restrict__ sc_tbl)
{
    const u32 lane = L.row * 16 + L.c;
    u32 top_col = 0;
#pragma unroll 1
    for (int i = 0; i < SC_ROUNDS; i++) {                  // every row fetch up front (the addresses depend on sigma alone)
        u64 cols = (u64)sigma_word(2 * i) | ((u64)sigma_word(2 * i + 1) << 32);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int c = 4 * i + 3 - j;
            if (c < SC_COLS) rowq[c * 64 + lane] = comb_limb(sc_tbl, L, (u32)cols & 0xffffu, by_row(L, 1, 0, 2, 2));
            if (c == SC_COLS - 1) top_col = (u32)cols & 0xffffu;
            cols >>= 16;
        }
    }
    const u32 ypx = comb_limb(sc_tbl, L, top_col, 0), ymx = comb_limb(sc_tbl, L, top_col, 1), t2d = comb_limb(sc_tbl, L, top_col, 2);
    const u32 two = L.c == 0 ? 2u : 0u;
    wave_fence();
    put_a(lds, L, L.row, L.upper ? (L.odd_row ? t2d : two) : (L.odd_row ? ypx + ymx : ypx + L.p2 - ymx));
    u32 v = mul_level(lds, L, L.row, by_row(L, SLOT_ONE, SLOT_ONE, SLOT_ONE, SLOT_KDI));   // (2x, 2y, 2, 2xy) of the top column's row
#pragma unroll 1
    for (int c = SC_COLS - 2; c >= 0; c--) {
        v = ge_dbl(lds, L, v);
        v = ge_add(lds, L, v, rowq[c * 64 + lane]);
    }
    return v;
}

// the point in the rows -> its precomputed form (Y+X, Y-X, 2dT, 2Z) as multiplier forms in slots `slot` .. `slot + 3` of the region
// `out` (another wave's, for the final sum); SLOT_K2D of `lds` must be set
C25519_DEV void store_pe(u32* lds, u32* out, const Lane& L, int slot, u32 v)
{
    u32 ev, od;
    pair_exchange(ev, od, v);                             // lower pair: X, Y; upper pair: Z, T
    const u32 val = L.upper ? (L.odd_row ? od : ev + ev) : (L.odd_row ? od + L.p2 - ev : ev + od);
    put_a(lds, L, L.row, val);                            // Y+X, Y-X, 2Z, T
    const u32 w = mul_level(lds, L, L.row, by_row(L, SLOT_ONE, SLOT_ONE, SLOT_ONE, SLOT_K2D));
    put_y(out, L, slot + by_row(L, 0, 1, 3, 2), w);       // fields: ypx, ymx, t2d, z2
}

// all-ones iff the point in the rows is the neutral element: X == 0 and Y == Z
C25519_DEV u32 is_neutral(u32* lds, const Lane& L, u32 v)
{
    put_a(lds, L, L.row, v);
    wave_fence();
    fe X, Y, Z, d;
    get_fe(X, lds, 0);
    get_fe(Y, lds, 1);
    get_fe(Z, lds, 2);
    wave_fence();
    fe_sub(d, Y, Z);
    u32 xw[8], dw[8], acc = 0;
    fe_to_words(xw, X);
    fe_to_words(dw, d);
#pragma unroll
    for (int i = 0; i < 8; i++) acc |= xw[i] | dw[i];
    return acc == 0 ? 0xffffffffu : 0u;
}

// ---- the two-phase API, one operation per wave ------------------------------------------------------------------------
// ed25519_Verify_Check for one (signature, message) pair in the REFERENCE's own operation order (edp_PolyPointMultiply,
// ed25519_verify.c:243-280: a context is caller storage and, for an off-curve key, the value of the sum depends on the order
// it is formed in): T = s*B + h*Q with Q's 16-row 4-fold table in LDS multiplier forms (slots QSLOT0 + row * 4 + field, unpacked from
// the context's canonical words) and the 8-fold base table's rows fetched from device memory before the walk starts.
constexpr int QSLOT0 = NSLOTS;
constexpr int Q_ROWQ_OFF = (QSLOT0 + 16 * 4) * SLOT_WORDS, Q_LDS_WORDS = Q_ROWQ_OFF + 32 * 64;

// limb L.c of the field an addition's first level multiplies THIS row by, row idx of the reference's table T (limb-major [30][256])
C25519_DEV u32 ref_limb_for_add(const u32* __restrict__ tbl, const Lane& L, u32 idx)
{
    const u32 f = by_row(L, 1, 0, 2, 2);
    return tbl[(f * 10 + (L.c < 10 ? L.c : 9)) * 256 + idx];
}

// s and h are consumed
C25519_DEV u32 poly_mult(u32* lds, const Lane& L, u32 (&s)[8], u32 (&h)[8], const u32* __restrict__ ref_tbl)
{
    const u32 lane = L.row * 16 + L.c;
#pragma unroll 1
    for (int i = 0; i < 32; i++) lds[Q_ROWQ_OFF + i * 64 + lane] = ref_limb_for_add(ref_tbl, L, fold8_next(s));
    u32 v;
    {   // S = row h_0 of Q's table as an extended point (ge_from_pe)
        const u32 m = fold4_next(h, false);
        const u32* row = lds + (QSLOT0 + m * 4) * SLOT_WORDS + YO_OFF + 10 + (L.c < 10 ? L.c : 9);
        const u32 a = row[0], b = row[SLOT_WORDS], t2d = row[2 * SLOT_WORDS], z2 = row[3 * SLOT_WORDS];
        wave_fence();
        put_a(lds, L, L.row, L.upper ? (L.odd_row ? t2d : z2) : (L.odd_row ? a + b : a + L.p2 - b));
        v = mul_level(lds, L, L.row, by_row(L, SLOT_ONE, SLOT_ONE, SLOT_ONE, SLOT_KDI));
    }
#pragma unroll 1
    for (int i = 1; i < 32; i++) {
        v = ge_dbl(lds, L, v);
        v = ge_add_pe(lds, L, v, QSLOT0 + fold4_next(h, false) * 4, 0u);
    }
#pragma unroll 1
    for (int i = 0; i < 32; i++) {
        v = ge_dbl(lds, L, v);
        v = ge_add(lds, L, v, lds[Q_ROWQ_OFF + i * 64 + lane]);
        v = ge_add_pe(lds, L, v, QSLOT0 + fold4_next(h, true) * 4, 0u);
    }
    return v;
}

// ed25519_Verify_Init for ONE key (ed25519_verify.c:179-232; ge25519.cuh: qtable_build): the 16-row 4-fold table of Q = -A
// -- row r = sum over the set bits i of r of 2^(64 i) Q, as (Y+X, Y-X, 2dT, 2Z) -- by the whole wave, in the reference's order
// (three times 64 doublings; row top + s = Q_blk + row s with Q_blk the extended and row s the precomputed operand), every row
// written to `rows_out` as four canonical 32-byte fields, byte for byte what the per-lane kernel writes (the one-key fast path
// compares a context with that).  xl, yl: limb L.c of Q's affine x and y.  Uses the slots QSLOT0 .. for the rows.
C25519_DEV void qtable_build_coop(u32* lds, const Lane& L, u32 xl, u32 yl, u32* __restrict__ rows_out /* 16 x 32 words */)
{
    const u32 one = L.c == 0 ? 1u : 0u;
    put_y(lds, L, SLOT_K2D, my_limb(lds, L, fe_const(K_2D)));
    // a row's precomputed form into its slots AND, canonical, into the context
    auto store_row = [&](int r, u32 v) {
        u32 ev, od;
        pair_exchange(ev, od, v);                         // lower pair: X, Y; upper pair: Z, T
        const u32 val = L.upper ? (L.odd_row ? od : ev + ev) : (L.odd_row ? od + L.p2 - ev : ev + od);
        put_a(lds, L, L.row, val);                        // Y+X, Y-X, 2Z, T
        const u32 w = mul_level(lds, L, L.row, by_row(L, SLOT_ONE, SLOT_ONE, SLOT_ONE, SLOT_K2D));
        put_y(lds, L, QSLOT0 + r * 4 + by_row(L, 0, 1, 3, 2), w);
        put_a(lds, L, by_row(L, 0, 1, 3, 2), w);          // fields in context order: ypx, ymx, t2d, z2
        wave_fence();
        const u32 lane = L.row * 16 + L.c;
        if (lane < 4) {                                   // lane f: field f, canonical
            fe t;
            get_fe(t, lds, lane);
            u32 w8[8];
            fe_to_words(w8, t);
            uint4* out = reinterpret_cast<uint4*>(rows_out + r * 32 + 8 * lane);
            out[0] = make_uint4(w8[0], w8[1], w8[2], w8[3]);
            out[1] = make_uint4(w8[4], w8[5], w8[6], w8[7]);
        }
        wave_fence();
    };
    // row 0: the neutral element (1, 1, 0, 2)
    {
        const u32 lane = L.row * 16 + L.c;
        put_y(lds, L, QSLOT0 + by_row(L, 0, 1, 3, 2), L.upper ? (L.odd_row ? 0u : one + one) : one);
        if (lane < 4) {
            uint4* out = reinterpret_cast<uint4*>(rows_out + 8 * lane);
            out[0] = make_uint4(lane == 2 ? 0u : lane == 3 ? 2u : 1u, 0, 0, 0);
            out[1] = make_uint4(0, 0, 0, 0);
        }
    }
    // Q = (x : y : 1 : x y)
    put_a(lds, L, L.row, L.upper ? (L.odd_row ? xl : one) : (L.odd_row ? yl : xl));
    put_y(lds, L, 4 + L.row, yl);
    u32 q = mul_level(lds, L, L.row, by_row(L, SLOT_ONE, SLOT_ONE, SLOT_ONE, 7));
    store_row(1, q);
#pragma unroll 1
    for (int blk = 1; blk < 4; blk++) {                   // Q <- 2^64 Q, then rows [2^blk, 2^(blk+1))
#pragma unroll 1
        for (int i = 0; i < 64; i++) q = ge_dbl(lds, L, q);
        const int top = 1 << blk;
        store_row(top, q);
#pragma unroll 1
        for (int s = 1; s < top; s++) store_row(top + s, ge_add_pe(lds, L, q, QSLOT0 + s * 4, 0u));
    }
}

// all-ones iff sigma*B + tau*Q + rho*Rn is the neutral element.  tq / tr: the element's packed window tables;
// sigma_w(w), tau_w(w), rho_w(w): words of its scalars; sc_tbl: the walk's comb table; top: first digit (>= SC_ROUNDS).
// TABLES_IN_LDS: the window tables' multiplier forms are in their slots already (wtable_build_lds), tq / tr are not read.
template <bool TABLES_IN_LDS = false, typename Words>
C25519_DEV u32 walk_is_neutral(u32* lds, const Lane& L, const Words& sc, const u32* __restrict__ tq, const u32* __restrict__ tr,
                               const u32* __restrict__ sc_tbl, int top)
{
    const u32 lane = L.row * 16 + L.c;
    // tables -> LDS multiplier forms
#pragma unroll 1
    for (int t = 0; t < (TABLES_IN_LDS ? 0 : 2); t++)
#pragma unroll 1
        for (int r = 0; r < WTABLE_ROWS; r++)
#pragma unroll
            for (int f = 0; f < 4; f++)
                put_y(lds, L, VSLOT0 + (t * WTABLE_ROWS + r) * 4 + f, packed_limb((t ? tr : tq) + r * ROW_WORDS + 8 * f, L));
    // sigma's comb rows, in the order the walk meets them: round i, step j -> column 4i + 3 - j
#pragma unroll 1
    for (int i = 0; i < SC_ROUNDS; i++) {
        u64 cols = (u64)sc.sigma_word(2 * i) | ((u64)sc.sigma_word(2 * i + 1) << 32);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            if (4 * i + 3 - j < SC_COLS) lds[V_ROWQ_OFF + (i * 4 + j) * 64 + lane] = comb_limb_for_add(sc_tbl, L, (u32)cols & 0xffffu);
            cols >>= 16;
        }
    }
    put_y(lds, L, SLOT_KDI, my_limb(lds, L, fe_const(K_DI)));
    auto digit = [&](u32 word, int i, u32& neg) -> u32 { return signed16_of(neg, word, i & 7); };
    u32 v;
    {   // the first digit: S = +-TQ[m] as an extended point (ge_from_pe), then += +-TR[m2]
        u32 neg, neg2;
        const u32 m = digit(sc.tau_word(top >> 3), top, neg), m2 = digit(sc.rho_word(top >> 3), top, neg2);
        const u32* row = lds + (VSLOT0 + m * 4) * SLOT_WORDS + YO_OFF + 10 + (L.c < 10 ? L.c : 9);      // plain limbs of a field
        const u32 a = row[(neg ? 1 : 0) * SLOT_WORDS], b = row[(neg ? 0 : 1) * SLOT_WORDS];             // Y+X, Y-X of +-row
        const u32 t2d = row[2 * SLOT_WORDS], z2 = row[3 * SLOT_WORDS];
        wave_fence();
        put_a(lds, L, L.row, L.upper ? (L.odd_row ? (neg ? L.p2 - t2d : t2d) : z2) : (L.odd_row ? a + b : a + L.p2 - b));
        v = mul_level(lds, L, L.row, by_row(L, SLOT_ONE, SLOT_ONE, SLOT_ONE, SLOT_KDI));                 // 2x, 2y, 2z, 2xy
        v = ge_add_pe(lds, L, v, VSLOT0 + (WTABLE_ROWS + m2) * 4, neg2);
    }
#pragma unroll 1
    for (int i = top - 1; i >= 0; i--) {
        if (i >= SC_ROUNDS) {
#pragma unroll 1
            for (int j = 0; j < 4; j++) v = ge_dbl(lds, L, v);
        } else {
#pragma unroll 1
            for (int j = 0; j < 4; j++) {
                v = ge_dbl(lds, L, v);
                if (4 * i + 3 - j < SC_COLS) v = ge_add(lds, L, v, lds[V_ROWQ_OFF + (i * 4 + j) * 64 + lane]);
            }
        }
        u32 neg;
        const u32 mq = digit(sc.tau_word(i >> 3), i, neg);
        v = ge_add_pe(lds, L, v, VSLOT0 + mq * 4, neg);
        const u32 mr = digit(sc.rho_word(i >> 3), i, neg);
        v = ge_add_pe(lds, L, v, VSLOT0 + (WTABLE_ROWS + mr) * 4, neg);
    }
    // neutral element: X == 0 and Y == Z
    put_a(lds, L, L.row, v);
    wave_fence();
    fe X, Y, Z, d;
    get_fe(X, lds, 0);
    get_fe(Y, lds, 1);
    get_fe(Z, lds, 2);
    wave_fence();
    fe_sub(d, Y, Z);
    u32 xw[8], dw[8], acc = 0;
    fe_to_words(xw, X);
    fe_to_words(dw, d);
#pragma unroll
    for (int i = 0; i < 8; i++) acc |= xw[i] | dw[i];
    return acc == 0 ? 0xffffffffu : 0u;
}

}  // namespace coop
}  // namespace c25519
