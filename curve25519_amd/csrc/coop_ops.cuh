// curve25519_amd/csrc/coop_ops.cuh -- what ONE WAVE does per operation: the bodies of the per-wave kernels of engine.hip
// (k_x25519_coop, k_ed25519_keypair_coop, ... -- what a call of a few elements runs, the reference's own single-call
// prototypes above all), the way lanes.cuh holds what one LANE does per operation for the batch kernels.  Ladder, walks,
// inversion and the last multiplications are cooperative (coop25519.cuh: a field element limb-per-lane, up to four products
// at a time); hashing, scalar arithmetic, the decoding of the inputs and the canonical encoding of the results are the batch
// kernels' per-lane code, run by every lane on the same values.  Kept apart from the kernels so that the CPU tests can run
// the same source as 64 lock-step lanes on the host (tests/host_emul/coop_wave.h).
#pragma once
#include "lanes.cuh"
#include "verify_fast.cuh"
#include "coop25519.cuh"

namespace c25519 {

// scratch of the lattice path of one verification call (engine.hip: verify_run carves it)
struct FastScratch {
    u32 *tables;            // per lane: window table of +-Q, then of -R (2 x WTABLE_WORDS of packed 128-byte rows, 128-byte aligned)
    u32 *sigma, *rho, *tau, *flags;
    u32 *slow_list;         // indices of the elements the reference-order kernel has to decide ...
    u32 *slow_count;        // ... and how many; [1], [2]: how many elements `order` holds from its front / from its back
    u32 *order;             // the walk's lane j takes element order[j]: elements whose scalars start at digit 32 or below
                            // from the front, the few longer ones from the back, so that a wave of 64 rarely holds one
    u32 *pflags;            // [2n], the quad path only (k_ed25519_verify_quad_prep): all-ones iff key e (R of e, at n + e) decoded onto the curve
    u32 *slow_report;       // a word that outlives the call's scratch: the count again, for c25519_amd_verify_last_slow_elements
    int lat_cap_bits;       // longest short vector the walk takes (LAT_CAP_BITS; lower only under the test knob VERIFY_LAT_CAP_BITS)
};
constexpr size_t FAST_TABLE_WORDS = 2 * WTABLE_WORDS;
constexpr u32 FLAG_R_OK = 1u, FLAG_KEY_OK = 2u, FLAG_FITS = 4u, FLAG_TAU_NEG = 8u, FLAG_SLOW = 16u;

namespace coop {

// the constant 1 in multiplier form: every operation's first step
C25519_DEV void setup_one(u32* lds, const Lane& L)
{
    fe one;
    fe_set_u32(one, 1);
    put_y(lds, L, SLOT_ONE, my_limb(lds, L, one));
}

// curve25519_dh_CreateSharedKey / CalculatePublicKey (curve25519_dh.c:94-157, 191-208) for element e.  lds: ROWQ_OFF words.
template <bool BASE9>
C25519_DEV void x25519_one(u32* lds, const Lane& L, void* out, const void* pk, void* sk, size_t e, const CallWords* cw = nullptr, const DoneWord* done = nullptr)
{
    u32 u[8] = { 9, 0, 0, 0, 0, 0, 0, 0 }, k[8];
    if (cw && cw->use) {                                 // a call of one: pk in words 0..7, sk in words 8..15 of the arguments
#pragma unroll
        for (int i = 0; i < 8; i++) { if (!BASE9) u[i] = cw->w[i]; k[i] = cw->w[8 + i]; }
    } else {
        if (!BASE9) load32(u, pk, e);
        load32(k, sk, e);
    }
    clamp_words(k);
    if (threadIdx.x == 0) store32(sk, e, k);            // the reference clamps in the caller's buffer
    fe X1, one;
    fe_from_words(X1, u);
    fe_set_u32(one, 1);
    const u32 x1 = my_limb(lds, L, X1), o1 = my_limb(lds, L, one);
    put_y(lds, L, SLOT_X1, x1);
    put_y(lds, L, SLOT_ONE, o1);
    // P = (X1 : 1) in rows 0, 1 and Q = 2P in rows 2, 3; bit 254 is the leading one (curve25519_dh.c:123-125 with zr = 1)
    u32 v = L.odd_row ? o1 : x1;
    {
        const u32 q = mont_double(lds, L, v);
        v = L.upper ? q : v;
    }
    u32 prev = 1;
#pragma unroll 1
    for (int w = 7; w >= 0; w--) {
        u32 kw = k[7];                                   // the scalar's words as a queue (x25519.cuh)
#pragma unroll
        for (int t = 7; t > 0; t--) k[t] = k[t - 1];
        const int top = (w == 7) ? 29 : 31, bottom = (w == 0) ? 3 : 0;
        kw <<= (31 - top);
#pragma unroll 1
        for (int b = top; b >= bottom; b--) {
            const u32 bit = kw >> 31;
            kw <<= 1;
#ifndef C25519_COOP_SKIP_LADDER                           // timing experiments only (tools/build_variants.sh): wrong results
            v = ladder_step<BASE9>(lds, L, v, (u32)0 - (u32)(bit == prev));
#endif
            prev = bit;
        }
    }
    // P = the sum if the last bit was one, else the double (curve25519_dh.c:148-150): into both row pairs, x in the even
    // rows, z in the odd; then the three clamped-away low bits -- three doublings of P
    u32 lo, hi;
    half_exchange(lo, hi, v);
    u32 p = hi ^ ((hi ^ lo) & ((u32)0 - prev));
#pragma unroll 1
    for (int i = 0; i < 3; i++) p = mont_double(lds, L, p);
    // x / z: the odd rows' inverse times the even rows' x, in every row; canonical bytes by every lane
#ifndef C25519_COOP_SKIP_INVERT
    const u32 zi = invert(lds, L, p);
#else
    const u32 zi = p;
#endif
    u32 px, pz, ix, iz;
    pair_exchange(px, pz, p);
    pair_exchange(ix, iz, zi);
    const u32 r = mul2(lds, L, px, iz);
    put_a(lds, L, L.row, r);
    wave_fence();
    fe R;
    get_fe(R, lds, 0);
    u32 wds[8];
    fe_to_words(wds, R);
    if (threadIdx.x == 0) store32(out, e, wds);         // written last: `out` may alias `pk`
    finish(lds, ROWQ_OFF, done);
}

// curve25519_dh_CreateSharedKey for element e by a workgroup of TWO waves (coop25519.cuh: the ladder step in two product levels,
// wave 0 the differential addition with x1 times the sum carried along, wave 1 the doubling): 510 levels and 255 barriers instead
// of 765 levels.  lds_all: X2_LDS_WORDS words.  The set-up
// and the tail (three doublings, inversion, canonical bytes) are x25519_one's, in a slot region of the wave's own.
C25519_DEV void x25519_two_waves(u32* lds_all, void* out, const void* pk, void* sk, size_t e, const CallWords* cw = nullptr, const DoneWord* done = nullptr)
{
    const int wave = threadIdx.x >> 6;
    const Lane L = make_lane(threadIdx.x & 63);
    u32* lds = lds_all + (X2_SHARED_SLOTS + wave * NSLOTS) * SLOT_WORDS;
    u32 u[8], k[8];
    if (cw && cw->use) {                                 // a call of one: the records came with the kernel's arguments
#pragma unroll
        for (int i = 0; i < 8; i++) { u[i] = cw->w[i]; k[i] = cw->w[8 + i]; }
    } else {
        load32(u, pk, e);
        load32(k, sk, e);
    }
    clamp_words(k);
    if (threadIdx.x == 0) store32(sk, e, k);            // the reference clamps in the caller's buffer
    fe X1, one;
    fe_from_words(X1, u);
    fe_set_u32(one, 1);
    const u32 x1 = my_limb(lds, L, X1), o1 = my_limb(lds, L, one);
    u32 v;
    if (wave == 0) {                                     // P = (X1 : 1) in rows 0, 1 and x1 P = (x1^2 : x1) in rows 2, 3
        const u32 xx = sqr_n(lds, L, x1, 1);
        v = L.upper ? (L.odd_row ? x1 : xx) : (L.odd_row ? o1 : x1);
    } else {                                             // Q = 2P, in both row pairs
        v = mont_double(lds, L, L.odd_row ? o1 : x1);
    }
    u32 prev = 1, par = 0;
#pragma unroll 1
    for (int w = 7; w >= 0; w--) {
        u32 kw = k[7];                                   // the scalar's words as a queue (x25519.cuh)
#pragma unroll
        for (int t = 7; t > 0; t--) k[t] = k[t - 1];
        const int top = (w == 7) ? 29 : 31, bottom = (w == 0) ? 3 : 0;
        kw <<= (31 - top);
#pragma unroll 1
        for (int b = top; b >= bottom; b--) {
            const u32 bit = kw >> 31;
            kw <<= 1;
            v = wave == 0 ? ladder2_step_sum(lds_all, lds, L, v, par) : ladder2_step_double(lds_all, lds, L, v, (u32)0 - (u32)(bit == prev), par);
            par ^= 1u;
            prev = bit;
        }
    }
    // P = the sum if the last bit was one, else the double (curve25519_dh.c:148-150): wave 1 hands the double over and is done
    const u32 slot = X2_PUB + 6 * par + 4 + (L.odd_row ? 1 : 0);
    if (wave != 0) put_a(lds_all, L, slot, v);
    __syncthreads();
    if (wave != 0) return;                               // (no barrier behind this point)
    u32 lo, hi;
    half_exchange(lo, hi, v);                            // lo: x, z of the sum in both row pairs
    const u32 dbl = lds_all[slot * SLOT_WORDS + A_OFF + (L.c < 10 ? L.c : 0)];
    u32 p = dbl ^ ((dbl ^ lo) & ((u32)0 - prev));
#pragma unroll 1
    for (int i = 0; i < 3; i++) p = mont_double(lds, L, p);
    const u32 zi = invert(lds, L, p);
    u32 px, pz, ix, iz;
    pair_exchange(px, pz, p);
    pair_exchange(ix, iz, zi);
    const u32 r = mul2(lds, L, px, iz);
    put_a(lds, L, L.row, r);
    wave_fence();
    fe R;
    get_fe(R, lds, 0);
    u32 wds[8];
    fe_to_words(wds, R);
    if (threadIdx.x == 0) store32(out, e, wds);         // written last: `out` may alias `pk`
    finish(lds_all, X2_LDS_WORDS, done);
}

// the fixed-base walk of the three operations below: over the wide comb (rows from device memory; a blinding context if given)
// or the eight LDS-comb tables read from device memory
template <bool WIDE>
C25519_DEV u32 base_mult_one(u32* lds, const Lane& L, const u32 (&k)[8], const u32* __restrict__ g_tbl, const u32* __restrict__ blind_ctx)
{
    if (WIDE) return ge_base_mult_wide(lds, L, k, g_tbl, blind_ctx);
    return ge_base_mult(lds, L, k, g_tbl);
}

// ed25519_CreateKeyPair (ed25519_sign.c:344-367) for element e.  lds: LDS_WORDS words.
template <bool WIDE, typename Sha = ShaPlain>
C25519_DEV void keypair_one(u32* lds, const Lane& L, void* pub, void* priv, const void* sk, size_t e, const u32* __restrict__ g_tbl,
                            const u32* __restrict__ blind_ctx, const DoneWord* done = nullptr, const Sha& sha = Sha())
{
    u32 seed[8], a[8], xw[8], yw[8], enc[8];
    u64 b_words[4];
    load32(seed, sk, e);
    ed_expand_seed(a, b_words, seed, sha);
    setup_one(lds, L);
    const u32 v = base_mult_one<WIDE>(lds, L, a, g_tbl, blind_ctx);
    ge_affine_words(xw, yw, lds, L, v);
    ge_pack(enc, xw, yw);
    if (threadIdx.x == 0) {
        store32(priv, 2 * e, seed);
        store32(priv, 2 * e + 1, enc);
        store32(pub, e, enc);
    }
    finish(lds, LDS_WORDS, done);
}

// curve25519_dh_CalculatePublicKey_fast (curve25519_dh.c:162-189): S = clamp(sk) * B on the Edwards side, u = (Z + Y) / (Z - Y)
template <bool WIDE>
C25519_DEV void public_fast_one(u32* lds, const Lane& L, void* pk, void* sk, size_t e, const u32* __restrict__ g_tbl, const DoneWord* done = nullptr)
{
    u32 k[8], wds[8];
    load32(k, sk, e);
    clamp_words(k);
    if (threadIdx.x == 0) store32(sk, e, k);
    setup_one(lds, L);
    const u32 v = base_mult_one<WIDE>(lds, L, k, g_tbl, nullptr);
    u32 ev, od, y, z, t;
    pair_exchange(ev, od, v);                             // lower pair: X, Y; upper pair: Z, T
    half_exchange(y, t, od);                              // y: Y in every row
    half_exchange(t, z, ev);                              // z: Z in every row
    const u32 zi = invert(lds, L, z + L.p2 - y);
    const u32 r = mul2(lds, L, z + y, zi);
    put_a(lds, L, L.row, r);
    wave_fence();
    fe R;
    get_fe(R, lds, 0);
    fe_to_words(wds, R);
    if (threadIdx.x == 0) store32(pk, e, wds);
    finish(lds, LDS_WORDS, done);
}

// ed25519_SignMessage (ed25519_sign.c:370-422) for element e
template <bool WIDE, typename Sha = ShaPlain>
C25519_DEV void sign_one(u32* lds, const Lane& L, void* sig, const void* priv, const Msgs& msgs, size_t e, const u32* __restrict__ g_tbl,
                         const u32* __restrict__ blind_ctx, const DoneWord* done = nullptr, const Sha& sha = Sha())
{
    u32 seed[8], pkw[8], a[8], r[8], xw[8], yw[8], enc[8], s[8];
    load32(seed, priv, 2 * e);
    load32(pkw, priv, 2 * e + 1);
    ed_sign_nonce(a, r, seed, msgs.ptr(e), msgs.len(e), sha);
    setup_one(lds, L);
    const u32 v = base_mult_one<WIDE>(lds, L, r, g_tbl, blind_ctx);
    ge_affine_words(xw, yw, lds, L, v);
    ge_pack(enc, xw, yw);
    ed_sign_s(s, enc, pkw, msgs.ptr(e), msgs.len(e), a, r, sha);
    if (threadIdx.x == 0) {
        store32(sig, 2 * e, enc);
        store32(sig, 2 * e + 1, s);
    }
    finish(lds, LDS_WORDS, done);
}

// ed25519_Blinding_Init (ed25519_sign.c:289-331) for one context: t * B over the wide comb and its affine conversion by the wave
C25519_DEV void blinding_init_one(u32* lds, const Lane& L, u32* ctx, const uint8_t* seed, size_t seed_len, const u32* __restrict__ wide,
                                  const DoneWord* done = nullptr)
{
    u32 t[8], bl[8], zr[8], xw[8], yw[8];
    ed_blinding_scalars(t, bl, zr, seed, seed_len);
    setup_one(lds, L);
    const u32 v = ge_base_mult_wide(lds, L, t, wide);
    ge_affine_words(xw, yw, lds, L, v);
    if (threadIdx.x == 0) ed_blinding_store(ctx, bl, zr, xw, yw);
    finish(lds, LDS_WORDS, done);
}

// ed25519_Verify_Init (ed25519_verify.c:179-232) for key e: the square root and the 16-row table by the whole wave.  lds: Q_LDS_WORDS words; rows: the context's 16 rows.
C25519_DEV void verify_init_one(u32* lds, const Lane& L, const void* pk, size_t e, u32* rows)
{
    u32 pkw[8];
    load32(pkw, pk, e);
    const u32 parity = pkw[7] >> 31;                        // -A: y as given (bit 255 stripped), x with the INVERTED parity, no
    pkw[7] &= 0x7fffffffu;                                  // validation (ed25519_verify.c:191-197; lanes.cuh: ed_decode_neg_key)
    fe Y;
    fe_from_words(Y, pkw);
    setup_one(lds, L);
    const u32 yl = my_limb(lds, L, Y);
    u32 xl, x_zero;
    (void)calc_x_checked(lds, L, xl, x_zero, yl, ~parity);
    qtable_build_coop(lds, L, xl, yl, rows);
}

// ed25519_Verify_Check (ed25519_verify.c:287-313) for pair e under the key of `ctx`: the reference's own operation order
// (poly_mult), one inversion, the comparison with enc(R).  ref_tbl: the reference's 256-row table.  lds: Q_LDS_WORDS words.
C25519_DEV void verify_check_one(u32* lds, const Lane& L, int* verdict, const void* sig, const u32* __restrict__ ctx, const Msgs& msgs,
                                 size_t e, const u32* __restrict__ ref_tbl)
{
    u32 pkw[8], Rw[8], Sw[8], h[8], xw[8], yw[8], enc[8];
#pragma unroll
    for (int j = 0; j < 8; j++) pkw[j] = ctx[j];
    load32(Rw, sig, 2 * e);
    ed_hram(h, Rw, pkw, msgs.ptr(e), msgs.len(e));
    sc_mod(h);
    load32(Sw, sig, 2 * e + 1);                            // raw 256 bits: no s < L check (ed25519_verify.c:308)
    setup_one(lds, L);
    put_y(lds, L, SLOT_KDI, my_limb(lds, L, fe_const(K_DI)));
#pragma unroll 1
    for (int r = 0; r < 16; r++)
#pragma unroll
        for (int f = 0; f < 4; f++) put_y(lds, L, QSLOT0 + r * 4 + f, packed_limb(ctx + 8 + r * 32 + 8 * f, L));
    const u32 v = poly_mult(lds, L, Sw, h, ref_tbl);
    ge_affine_words(xw, yw, lds, L, v);
    ge_pack(enc, xw, yw);
    u32 diff = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) diff |= enc[j] ^ Rw[j];
    if (threadIdx.x == 0) verdict[e] = diff == 0 ? 1 : 0;
}

// The whole lattice path of ONE element by a workgroup of THREE waves (k_ed25519_verify_one_per_group):
//   phase 1   wave 0 hashes and reduces (every lane on the same values) WHILE wave 1 decodes R and wave 2 the key, each square
//             root by the whole wave (calc_x_checked; they do not need the scalars);
//   phase 2   the equation sigma*B + tau*Q + rho*(-R) = O is three independent products, one wave each, every wave in an LDS
//             region of its own: wave 0 builds the key's window table with the whole wave (wtable_build_lds: straight into
//             the forms the walk reads, no round trip through memory) and walks tau over it, wave 1 does the same for R and rho,
//             wave 2 runs sigma*B over the comb by Horner's rule -- each needs the ~129 doublings the joint walk shared, but
//             side by side on three SIMDs: 330 product levels in a row instead of 440;
//   phase 3   waves 1 and 2 hand their points to wave 0 in precomputed form; two additions and the neutral-element test.
// Elements the path cannot decide go on the slow list exactly as in the batch kernels.
// LDS of the workgroup: wave 0: operand slots, one window table, eight hand-over slots; wave 1: operand slots, one table;
// wave 2: operand slots, the comb's row queue -- 45 KiB, three workgroups per CU
constexpr int V3_TABLE_SLOTS = WTABLE_ROWS * 4, V3_HANDOVER = VSLOT0 + V3_TABLE_SLOTS;
constexpr int V3_BASE1 = (V3_HANDOVER + 8) * SLOT_WORDS, V3_BASE2 = V3_BASE1 + (VSLOT0 + V3_TABLE_SLOTS) * SLOT_WORDS;
constexpr int V3_ROWQ2 = NSLOTS * SLOT_WORDS;
constexpr int V3_LDS_WORDS = V3_BASE2 + V3_ROWQ2 + SC_ROUNDS * 4 * 64;

// lds_all: V3_LDS_WORDS words; park: 40 words (limbs of the key's x, y and of R's); hand: 4 words (tau < 0; wave 0's flag bits;
// key on the curve; R decodes canonically)
C25519_DEV void verify_three_waves(u32* lds_all, u32* park, u32* hand, const FastScratch& fs, int* verdict, const void* sig, const void* pk,
                                   const Msgs& msgs, size_t n, size_t e, const u32* __restrict__ g_tbl)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    u32* lds = lds_all + (wave == 0 ? 0 : wave == 1 ? V3_BASE1 : V3_BASE2);
    if (wave == 0) {
        u32 pkw[8], Rw[8], Sw[8], cols[SIGMA_WORDS], rho[5], tau[5], tau_neg;
        load32(pkw, pk, e);
        load32(Rw, sig, 2 * e);
        load32(Sw, sig, 2 * e + 1);
        const u32 lat_ok = ed_verify_fast_scalars(cols, rho, tau, tau_neg, pkw, Rw, Sw, msgs.ptr(e), msgs.len(e), fs.lat_cap_bits);
        if (lane == 0) {
#pragma unroll
            for (int w = 0; w < SIGMA_WORDS; w++) fs.sigma[(size_t)w * n + e] = cols[w];
#pragma unroll
            for (int w = 0; w < 5; w++) { fs.rho[(size_t)w * n + e] = rho[w]; fs.tau[(size_t)w * n + e] = tau[w]; }
            const int top = lat_ok ? walk_top_digit(tau, rho) : 0;
            hand[0] = tau_neg;
            hand[1] = (lat_ok & FLAG_FITS) | (tau_neg & FLAG_TAU_NEG) | ((u32)top << 8);
        }
    } else {
        // ed_verify_fast_decode (verify_fast.cuh) by a whole wave each: wave 1 takes R -- which must be the canonical encoding of a
        // curve point, and is negated: the walk adds rho * (-R) --, wave 2 the key, decoded as -A exactly as ed25519_Verify_Init
        // does (the sign of tau turns it once more in phase 2)
        const u32 is_r = wave == 1 ? 0xffffffffu : 0u;
        const Lane L = make_lane(lane);
        u32 w[8], yw[8], cw[8];
        if (is_r) load32(w, sig, 2 * e); else load32(w, pk, e);
#pragma unroll
        for (int i = 0; i < 8; i++) yw[i] = w[i];
        const u32 sign = yw[7] >> 31;
        yw[7] &= 0x7fffffffu;
        fe Y;
        fe_from_words(Y, yw);
        const u32 parity = sign ^ (~is_r & 1u);
        setup_one(lds, L);
        const u32 yl = my_limb(lds, L, Y);
        u32 xl, x_zero;
        u32 ok = calc_x_checked(lds, L, xl, x_zero, yl, parity);
        fe_to_words(cw, Y);
        u32 diff = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) diff |= cw[i] ^ yw[i];
        // y < p, and the sign bit an encoder would have produced: x = 0 has parity 0 only
        const u32 canonical = (diff == 0 && !(x_zero && (parity & 1u))) ? 0xffffffffu : 0u;
        ok &= ~is_r | canonical;
        const u32 neg = carry_small(L, (u64)(L.p2 - xl));
        if (is_r) xl = neg;
        if (lane == 0) hand[is_r ? 3 : 2] = ok ? 1u : 0u;
        if (L.row == 0 && L.c < 10) {
            park[(is_r ? 20 : 0) + L.c] = xl;
            park[(is_r ? 30 : 10) + L.c] = yl;
        }
    }
    __syncthreads();
    const u32 f = hand[1] | (hand[2] ? FLAG_KEY_OK : 0u) | (hand[3] ? FLAG_R_OK : 0u);
    if ((f & (FLAG_KEY_OK | FLAG_FITS)) != (FLAG_KEY_OK | FLAG_FITS)) {      // off-curve key / over-long vector: the slow list
        if (threadIdx.x == 0) {
            fs.flags[e] = f | FLAG_SLOW;
            fs.slow_list[atomicAdd(fs.slow_count, 1u)] = (u32)e;
        }
        return;
    }
    if (threadIdx.x == 0) fs.flags[e] = f;
    const Lane L = make_lane(lane);
    const int top = (int)((f >> 8) & 63u);
    const u32 c = L.c < 10 ? L.c : 0;
    setup_one(lds, L);
    put_y(lds, L, SLOT_KDI, my_limb(lds, L, fe_const(K_DI)));
    u32 v;
    if (wave == 0) {                                        // |tau| * (+-Q)
        const u32 xl = hand[0] ? L.p2 - park[c] : park[c]; // tau < 0: the table of -Q (ed_verify_fast_decode)
        wtable_build_lds(lds, L, 0, xl, park[10 + c]);
        v = walk_point(lds, L, [&](int w) -> u32 { return fs.tau[(size_t)w * n + e]; }, VSLOT0, top);
    } else if (wave == 1) {                                 // rho * (-R)
        wtable_build_lds(lds, L, 0, park[20 + c], park[30 + c]);
        v = walk_point(lds, L, [&](int w) -> u32 { return fs.rho[(size_t)w * n + e]; }, VSLOT0, top);
        store_pe(lds, lds_all, L, V3_HANDOVER, v);
    } else {                                                // sigma * B
        put_y(lds, L, SLOT_K2D, my_limb(lds, L, fe_const(K_2D)));
        v = walk_comb(lds, lds + V3_ROWQ2, L, [&](int w) -> u32 { return fs.sigma[(size_t)w * n + e]; }, g_tbl + SC_TBL_OFFSET);
        store_pe(lds, lds_all, L, V3_HANDOVER + 4, v);
    }
    __syncthreads();
    if (wave != 0) return;
    v = ge_add_pe(lds, L, v, V3_HANDOVER, 0u);
    v = ge_add_pe(lds, L, v, V3_HANDOVER + 4, 0u);
    const u32 neutral = is_neutral(lds, L, v);
    if (lane == 0) verdict[e] = (neutral & f & FLAG_R_OK) ? 1 : 0;
}

}  // namespace coop
}  // namespace c25519
