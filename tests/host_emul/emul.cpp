// tests/host_emul/emul.cpp -- TEST INFRASTRUCTURE.  Drives the device headers of curve25519_amd/csrc (compiled for
// the host against the C model of the gfx950 primitives, valu_model.h) one lane at a time, the way the kernels of
// engine.hip drive them per lane.  Exposes a plain C interface for tests/test_host_emul.py.  The point is to test the
// DEVICE SOURCE (field / scalar / group / hashing / lane logic) on a CPU-only machine; the GPU tests then only have
// to establish that the kernels' indexing, LDS staging and scratch plumbing around these functions are right.
// Not part of the product and not a fallback: libcurve25519_amd.so contains no host arithmetic.
#define EMUL_COOP_WAVE_IMPL 1              // this translation unit holds the lock-step lane scheduler (coop_wave.h)
#include "coop_wave.h"
#include "lanes.cuh"
#include "verify_fast.cuh"
#include "coop_ops.cuh"
#include "quad25519.cuh"

#include <mutex>
#include <thread>
#include <vector>

using namespace c25519;

namespace c25519 { unsigned long long emul_mad_overflows = 0, emul_mad_count = 0; LatCounters emul_lat_counters = { 0, 0, 0 }; }
thread_local EmulWave* emul_wave = nullptr;
thread_local emul_dim3 emul_tid = { 0, 0, 0 };

namespace {

std::vector<u32> g_tbl;                 // [BASE_NT][30][128] signed comb tables, [30][256] the reference's table, [30][SC_ROWS] the walk's comb:
                                        // limb-major, as k_gen_base_table lays them out
std::vector<u32> g_tbl_bytes;           // [256][24]
std::once_flag g_tbl_once;

void build_tables()
{
    g_tbl.assign((size_t)ALL_TBL_WORDS, 0);
    g_tbl_bytes.assign(256 * 24, 0);
    for (int group = 0; group < BASE_NT; group++)
        for (u32 idx = 0; idx < (u32)BASE_ROWS; idx++) {
            u32 rows[3][8];
            ge_signed_comb_row(rows, idx, (BASE_NT - 1 - group) * BASE_STEP);
            for (int f = 0; f < 3; f++) {
                fe c;
                fe_from_words(c, rows[f]);
                for (int l = 0; l < 10; l++) g_tbl[(size_t)group * BASE_TBL_WORDS + (10 * f + l) * BASE_ROWS + idx] = c.v[l];
            }
        }
    for (u32 idx = 0; idx < (u32)SC_ROWS; idx++) {          // the lattice walk's signed comb
        u32 rows[3][8];
        ge_signed_comb_row(rows, idx, 0, SC_TEETH, SC_COLS);
        for (int f = 0; f < 3; f++) {
            fe c;
            fe_from_words(c, rows[f]);
            for (int l = 0; l < 10; l++) g_tbl[(size_t)SC_TBL_OFFSET + (10 * f + l) * SC_ROWS + idx] = c.v[l];
        }
    }
    for (u32 k = 0; k < 256; k++) {
        u32 rows[3][8];
        ge_base_table_row(rows, k, 0);
        for (int f = 0; f < 3; f++) {
            fe c;
            fe_from_words(c, rows[f]);
            for (int l = 0; l < 10; l++) g_tbl[(size_t)REF_TBL_OFFSET + (10 * f + l) * 256 + k] = c.v[l];
            for (int j = 0; j < 8; j++) g_tbl_bytes[k * 24 + 8 * f + j] = rows[f][j];
        }
    }
}

const u32* tables()
{
    std::call_once(g_tbl_once, build_tables);
    return g_tbl.data();
}

void rd32(u32 (&w)[8], const unsigned char* p, size_t i) { memcpy(w, p + 32 * i, 32); }
void wr32(unsigned char* p, size_t i, const u32 (&w)[8]) { memcpy(p + 32 * i, w, 32); }

void affine_pack_host(u32 (&enc)[8], const ge_ext& S)
{
    u32 xw[8], yw[8];
    ge_to_affine_words(xw, yw, S);
    ge_pack(enc, xw, yw);
}

// ---- the wide fixed-base comb (ge25519.cuh: WB_*): the same packed tables k_gen_wide_table writes, built here with one
// addition per row (Gray-code order: consecutive rows differ in one tooth, S +- 2 * 2^(20 j + extra) B) and ONE shared
// inversion per table instead of 250 doublings and an inversion per row; spot-checked against ge_signed_comb_row, the
// function the device generates its rows with, by emul_wide_row_check (tests/test_host_emul.py).
std::vector<u32> g_wide;
std::once_flag g_wide_once;
int g_base_comb = 0;                    // emul_set_base_comb: 1 = the fixed-base operations below walk the wide comb

// the four packed tables of the wide comb of ANY point given in affine precomputed form
void build_wide_of(std::vector<u32>& g_wide, const ge_pa& b)
{
    g_wide.assign(WB_TBL_WORDS, 0);
    for (int table = 0; table < WB_NT; table++) {
        const int extra = (WB_NT - 1 - table) * WB_STEP;
        // P[j] = 2^(20 j + extra) B, j = 0 .. 12
        ge_ext P[WB_TEETH];
        {
            ge_from_pa(P[0], b);
            for (int j = 0; j < extra; j++) ge_double(P[0]);
            for (int j = 1; j < WB_TEETH; j++) {
                P[j] = P[j - 1];
                for (int d = 0; d < WB_COLS; d++) ge_double(P[j]);
            }
        }
        ge_pe plus2[WB_TEETH - 1], minus2[WB_TEETH - 1], minus1[WB_TEETH - 1];
        for (int j = 0; j < WB_TEETH - 1; j++) {
            ge_ext D = P[j];
            ge_double(D);
            ge_to_pe(plus2[j], D);
            minus2[j] = plus2[j];
            std::swap(minus2[j].ypx, minus2[j].ymx);
            { fe t; fe_neg(t, plus2[j].t2d); fe_carry32(minus2[j].t2d, t); }
            ge_to_pe(minus1[j], P[j]);
            std::swap(minus1[j].ypx, minus1[j].ymx);
            { fe t; fe_neg(t, minus1[j].t2d); fe_carry32(minus1[j].t2d, t); }
        }
        std::vector<ge_ext> pts(WB_ROWS);
        ge_ext S = P[WB_TEETH - 1];                                   // row 0: the top tooth plus, every other tooth minus
        for (int j = 0; j < WB_TEETH - 1; j++) { ge_ext R; ge_add_pe(R, S, minus1[j]); S = R; }
        u32 gray = 0;
        pts[0] = S;
        for (u32 i = 1; i < (u32)WB_ROWS; i++) {
            const u32 g = i ^ (i >> 1), flip = g ^ gray;
            const int j = __builtin_ctz(flip);
            ge_ext R;
            ge_add_pe(R, S, (g & flip) ? plus2[j] : minus2[j]);
            S = R; gray = g;
            pts[g] = S;
        }
        // one inversion for the table (Montgomery's trick over Z)
        std::vector<fe> pre(WB_ROWS);
        fe acc = pts[0].Z;
        for (int i = 1; i < WB_ROWS; i++) { pre[i] = acc; fe_mul(acc, acc, pts[i].Z); }
        fe inv;
        fe_invert(inv, acc);
        for (int i = WB_ROWS - 1; i >= 0; i--) {
            fe zi;
            if (i) { fe_mul(zi, inv, pre[i]); fe_mul(inv, inv, pts[i].Z); } else zi = inv;
            fe x, y, t, row[3];
            fe_mul(x, pts[i].X, zi);
            fe_mul(y, pts[i].Y, zi);
            fe_add(row[0], y, x);
            fe_sub(row[1], y, x);
            fe_mul(t, x, y);
            fe_mul(row[2], t, fe_const(K_2D));
            u32* out = g_wide.data() + ((size_t)table * WB_ROWS + i) * WB_ROW_WORDS;
            for (int f = 0; f < 3; f++) { u32 w[8]; fe_to_words(w, row[f]); memcpy(out + 8 * f, w, 32); }
            out[24] = 2;                                             // the fourth field: 2Z of an affine point (k_gen_wide_table)
        }
    }
}

void build_wide()
{
    u32 rows[3][8];
    ge_base_table_row(rows, 1, 0);                                   // B as (y+x, y-x, 2dxy)
    ge_pa b;
    fe_from_words(b.ypx, rows[0]); fe_from_words(b.ymx, rows[1]); fe_from_words(b.t2d, rows[2]);
    build_wide_of(g_wide, b);
}

const u32* wide_tables()
{
    std::call_once(g_wide_once, build_wide);
    return g_wide.data();
}

}  // namespace

extern "C" {

unsigned long long emul_mad_overflow_count(void) { return emul_mad_overflows; }
// v_mad_u64_u32 instructions issued since the last call (valu_model.h: mad64)
unsigned long long emul_mad_count_take(void) { return __atomic_exchange_n(&emul_mad_count, 0ULL, __ATOMIC_RELAXED); }

static void emul_fe_op_quad(unsigned char* out, const unsigned char* a, const unsigned char* b, size_t n, int op);
void emul_fe_op(unsigned char* out, const unsigned char* a, const unsigned char* b, size_t n, int op)
{
    if (op == 14) return emul_fe_op_quad(out, a, b, n, op);        // four lock-step lanes per record (k_fe_selftest_quad)
    for (size_t i = 0; i < n; i++) {
        u32 aw[8], bw[8], ow[8];
        rd32(aw, a, i); rd32(bw, b, i);
        fe_selftest_op(ow, aw, bw, op);
        wr32(out, i, ow);
    }
}

void emul_sc_op(unsigned char* out, const unsigned char* a, const unsigned char* b, size_t n, int op)
{
    for (size_t i = 0; i < n; i++) {
        u32 aw[16], bw[8], ow[8];
        memcpy(aw, a + 64 * i, 64); rd32(bw, b, i);
        sc_selftest_op(ow, aw, bw, op);
        wr32(out, i, ow);
    }
}

void emul_fold(unsigned char* out, const unsigned char* k, size_t n)
{
    for (size_t i = 0; i < n; i++) {
        u32 kw[8];
        rd32(kw, k, i);
        fold_selftest_op(out + 128 * i, kw);
    }
}

void emul_base_table(unsigned char* out /* 256 x 96 */)
{
    tables();
    memcpy(out, g_tbl_bytes.data(), 256 * 96);
}

// curve25519_dh_CreateSharedKey / CalculatePublicKey (pk == NULL) per element, sk clamped in place
void emul_x25519(unsigned char* out, const unsigned char* pk, unsigned char* sk, size_t n)
{
    for (size_t i = 0; i < n; i++) {
        u32 u[8] = { 9, 0, 0, 0, 0, 0, 0, 0 }, k[8], w[8];
        if (pk) rd32(u, pk, i);
        rd32(k, sk, i);
        clamp_words(k);
        wr32(sk, i, k);
        fe PX, PZ, zi;
        if (pk) x25519_ladder_xz<false>(PX, PZ, u, k);
        else x25519_ladder_xz<true>(PX, PZ, u, k);
        fe_invert(zi, PZ);
        fe_mul(PX, PX, zi);
        fe_to_words(w, PX);
        wr32(out, i, w);
    }
}

static void base_mult_any(ge_ext& S, const u32 (&k)[8], const unsigned char* blinding);

// 0: the fixed-base operations walk the 8 x 32 comb (the LDS tables), 1: the wide 13 x 20 comb (tunable BASE_COMB)
void emul_set_base_comb(int wide) { g_base_comb = wide; }

// rows `idx[i]` of wide table `table` as the device generates them (ge_signed_comb_row) against the table built here:
// number of differing rows
int emul_wide_row_check(int table, const unsigned* idx, size_t n)
{
    int bad = 0;
    for (size_t i = 0; i < n; i++) {
        u32 rows[3][8];
        ge_signed_comb_row(rows, idx[i] % WB_ROWS, (WB_NT - 1 - table) * WB_STEP, WB_TEETH, WB_COLS);
        const u32* have = wide_tables() + ((size_t)table * WB_ROWS + idx[i] % WB_ROWS) * WB_ROW_WORDS;
        if (memcmp(rows, have, 96) != 0) bad++;
    }
    return bad;
}

void emul_x25519_public_fast(unsigned char* pk, unsigned char* sk, size_t n)
{
    for (size_t i = 0; i < n; i++) {
        u32 k[8], w[8];
        rd32(k, sk, i);
        clamp_words(k);
        wr32(sk, i, k);
        ge_ext S;
        base_mult_any(S, k, nullptr);
        fe num, den, t;
        fe_add(t, S.Z, S.Y);  fe_carry32(num, t);
        fe_sub(t, S.Z, S.Y);  fe_carry32(den, t);
        fe_invert(t, den);
        fe_mul(num, num, t);
        fe_to_words(w, num);
        wr32(pk, i, w);
    }
}

void emul_blinding_init(unsigned char* ctx /* 192 */, const unsigned char* seed, size_t len)
{
    u32 w[BLIND_WORDS];
    ed_blinding_init_lane(w, seed, len, tables());
    memcpy(ctx, w, sizeof w);
}

static void base_mult_wide(ge_ext& S, const u32 (&k)[8], const fe* zr, bool final_t)
{
    unsigned short cols[WB_COLS];
    wb_columns(cols, 1, k);
    if (final_t) ge_base_mult_wide<true>(S, wide_tables(), cols, 1, zr);
    else ge_base_mult_wide<false>(S, wide_tables(), cols, 1, zr);
}

static void base_mult_any(ge_ext& S, const u32 (&k)[8], const unsigned char* blinding)
{
    if (g_base_comb == 1) {
        if (blinding) {
            u32 w[BLIND_WORDS];
            memcpy(w, blinding, sizeof w);
            ge_base_mult_blinded_with(S, k, w, [&](ge_ext& P, const u32 (&t)[8], const fe& zr) { base_mult_wide(P, t, &zr, true); });
        } else {
            base_mult_wide(S, k, nullptr, false);
        }
        return;
    }
    if (blinding) {
        u32 w[BLIND_WORDS];
        memcpy(w, blinding, sizeof w);
        ge_base_mult_blinded(S, k, w, tables());
    } else {
        ge_base_mult(S, k, tables());
    }
}

void emul_ed25519_keypair(unsigned char* pub, unsigned char* priv, const unsigned char* blinding,
                          const unsigned char* sk, size_t n)
{
    for (size_t i = 0; i < n; i++) {
        u32 seed[8], a[8], enc[8];
        u64 b_words[4];
        rd32(seed, sk, i);
        ed_expand_seed(a, b_words, seed);
        ge_ext S;
        base_mult_any(S, a, blinding);
        affine_pack_host(enc, S);
        wr32(priv, 2 * i, seed);
        wr32(priv, 2 * i + 1, enc);
        wr32(pub, i, enc);
    }
}

void emul_ed25519_sign(unsigned char* sig, const unsigned char* priv, const unsigned char* blinding,
                       const unsigned char* msg, size_t len, size_t n)
{
    for (size_t i = 0; i < n; i++) {
        u32 seed[8], pkw[8], a[8], r[8], enc[8], s[8];
        rd32(seed, priv, 2 * i);
        rd32(pkw, priv, 2 * i + 1);
        ed_sign_nonce(a, r, seed, msg + len * i, len);
        ge_ext S;
        base_mult_any(S, r, blinding);
        affine_pack_host(enc, S);
        ed_sign_s(s, enc, pkw, msg + len * i, len, a, r);
        wr32(sig, 2 * i, enc);
        wr32(sig, 2 * i + 1, s);
    }
}

// enc(T) of T = s*B + h*(-A) per element (what Verify_Check compares with enc(R)), and the verdicts
void emul_ed25519_verify(int* verdict, unsigned char* point /* may be NULL */, const unsigned char* sig,
                         const unsigned char* pk, const unsigned char* msg, size_t len, size_t n)
{
    const u32* tbl = tables() + (size_t)REF_TBL_OFFSET;
    std::vector<u32> q(QTABLE_LIMB_WORDS);
    for (size_t i = 0; i < n; i++) {
        u32 pkw[8], Rw[8], Sw[8], h[8], enc[8];
        rd32(pkw, pk, i);
        ge_ext Q, T;
        ed_decode_neg_key(Q, pkw);
        const QTableLimbs t{ q.data() };
        qtable_build(t, Q);
        rd32(Rw, sig, 2 * i);
        rd32(Sw, sig, 2 * i + 1);
        ed_hram(h, Rw, pkw, msg + len * i, len);
        sc_mod(h);
        ge_poly_mult(T, Sw, h, t, tbl);
        // the batched inversion maps Z == 0 to 0, which is also what z^(p-2) gives
        affine_pack_host(enc, T);
        if (point) wr32(point, i, enc);
        u32 diff = 0;
        for (int j = 0; j < 8; j++) diff |= enc[j] ^ Rw[j];
        if (verdict) verdict[i] = diff == 0;
    }
}

// the slow list's path (ed_verify_reference_order, verify_fast.cuh: the streamed table build and 4-fold walk) per element
void emul_ed25519_verify_slow(int* verdict, unsigned char* point, const unsigned char* sig, const unsigned char* pk,
                              const unsigned char* msg, size_t len, size_t n)
{
    const u32* tbl = tables() + (size_t)REF_TBL_OFFSET;
    std::vector<u32> q(QTABLE_LIMB_WORDS);
    for (size_t i = 0; i < n; i++) {
        u32 pkw[8], Rw[8], Sw[8];
        rd32(pkw, pk, i);
        rd32(Rw, sig, 2 * i);
        rd32(Sw, sig, 2 * i + 1);
        u32 enc[8];
        verdict[i] = ed_verify_reference_order(pkw, Rw, Sw, msg + len * i, len, q.data(), tbl, enc);
        wr32(point, i, enc);
    }
}

// the lattice fast path (verify_fast.cuh), one element at a time: verdicts (meaningful where need_slow[i] == 0) and the
// need_slow flags
void emul_ed25519_verify_fast_at(int* verdict, int* need_slow, const unsigned char* sig, const unsigned char* pk,
                                 const unsigned char* msg, size_t len, size_t n, int wave_top);
void emul_ed25519_verify_fast(int* verdict, int* need_slow, const unsigned char* sig, const unsigned char* pk,
                              const unsigned char* msg, size_t len, size_t n)
{
    emul_ed25519_verify_fast_at(verdict, need_slow, sig, pk, msg, len, n, 0);
}

// ... wave_top: the walk starts at max(the element's own first digit, wave_top) -- on the device a wave walks from its
// LONGEST element's first digit (k_ed25519_verify_fast_walk), the others' digits above their own being zero
void emul_ed25519_verify_fast_at(int* verdict, int* need_slow, const unsigned char* sig, const unsigned char* pk,
                                 const unsigned char* msg, size_t len, size_t n, int wave_top)
{
    const u32* tbl = tables() + (size_t)SC_TBL_OFFSET;
    std::vector<u32> q(2 * WTABLE_WORDS);
    for (size_t i = 0; i < n; i++) {
        u32 pkw[8], Rw[8], Sw[8], cols[SIGMA_WORDS], rho[5], tau[5], tau_neg;
        rd32(pkw, pk, i);
        rd32(Rw, sig, 2 * i);
        rd32(Sw, sig, 2 * i + 1);
        u32* const tq = q.data();
        u32* const tr = q.data() + WTABLE_WORDS;
        // the kernels' chain: scalars -> decode (key, then R) -> tables -> walk
        const u32 lat_ok = ed_verify_fast_scalars(cols, rho, tau, tau_neg, pkw, Rw, Sw, msg + len * i, len);
        fe QX, QY, RX, RY;
        const u32 q_ok = ed_verify_fast_decode(QX, QY, pkw, 0u, tau_neg);
        const u32 r_ok = ed_verify_fast_decode(RX, RY, Rw, 0xffffffffu, tau_neg);
        need_slow[i] = (lat_ok && q_ok) ? 0 : 1;
        verdict[i] = 0;
        if (need_slow[i]) continue;                                   // the walk's lanes skip listed elements
        wtable_build(tq, QX, QY);
        wtable_build(tr, RX, RY);
        int top = walk_top_digit(tau, rho);                          // a "wave" of one lane: every start digit gets exercised
        if (top < wave_top) top = wave_top;
        const WalkScalars sc{ cols, tau, rho, 1, 0 };
        const u32 neutral = ge_walk_is_neutral(sc, tq, tr, tbl, top < 8 ? 8 : top);
        verdict[i] = (r_ok && neutral) ? 1 : 0;
    }
}

// sc_comb_columns: sigma (n x 32 bytes, < L) -> SIGMA_WORDS words of 16-bit columns in walk order; dims = {teeth, columns}
void emul_comb_columns(unsigned* words_out, int* dims, const unsigned char* sigma, size_t n)
{
    dims[0] = SC_TEETH; dims[1] = SC_COLS; dims[2] = SIGMA_WORDS;
    for (size_t i = 0; i < n; i++) {
        u32 k[8], cols[SIGMA_WORDS];
        rd32(k, sigma, i);
        sc_comb_columns(cols, k);
        for (int w = 0; w < SIGMA_WORDS; w++) words_out[(size_t)SIGMA_WORDS * i + w] = cols[w];
    }
}

// fe_pack_words / fe_from_words on raw limbs (10 u32 per element in, 8 words out, 10 limbs back)
void emul_fe_pack_roundtrip(unsigned* words_out /* n x 8 */, unsigned* limbs_out /* n x 10 */, const unsigned* limbs_in /* n x 10 */, size_t n)
{
    for (size_t i = 0; i < n; i++) {
        fe a, b;
        for (int j = 0; j < 10; j++) a.v[j] = limbs_in[10 * i + j];
        u32 w[8];
        fe_pack_words(w, a);
        fe_from_words(b, w);
        for (int j = 0; j < 8; j++) words_out[8 * i + j] = w[j];
        for (int j = 0; j < 10; j++) limbs_out[10 * i + j] = b.v[j];
    }
}

// loop trips of the lattice reduction since the last call (outer Lehmer steps, their inner iterations, exact steps)
void emul_lattice_counters(unsigned long long* out3)
{
    out3[0] = emul_lat_counters.lehmer_outer; out3[1] = emul_lat_counters.lehmer_inner; out3[2] = emul_lat_counters.exact_steps;
    emul_lat_counters = LatCounters{ 0, 0, 0 };
}

// lattice reduction alone: h (n x 32) -> rho, tau (n x 20 bytes each), sign of tau, fits
void emul_lattice(unsigned char* rho_out, unsigned char* tau_out, int* tau_neg, int* fits, const unsigned char* h, size_t n)
{
    for (size_t i = 0; i < n; i++) {
        u32 hw[8], rho[5], tau[5], neg;
        rd32(hw, h, i);
        fits[i] = sc_lattice_short(rho, tau, neg, hw) ? 1 : 0;
        tau_neg[i] = neg ? 1 : 0;
        memcpy(rho_out + 20 * i, rho, 20);
        memcpy(tau_out + 20 * i, tau, 20);
    }
}

// lattice reduction with the elements grouped into "waves" of `lanes` host threads running in lock-step (valu_model.h):
// the wave-level control flow of sc_lattice_short -- __any-guarded loop exits, lanes that idle while others iterate -- as
// on the device.  Results must equal the one-lane run's.  n is rounded down to whole waves.
void emul_lattice_waves(unsigned char* rho_out, unsigned char* tau_out, int* tau_neg, int* fits, const unsigned char* h, size_t n,
                        int lanes)
{
    for (size_t base = 0; base + lanes <= n; base += lanes) {
        EmulWave w;
        w.lanes = lanes; w.arrived = 0; w.generation = 0; w.acc = false; w.result = false;
        pthread_mutex_init(&w.mu, nullptr);
        pthread_cond_init(&w.cv, nullptr);
        std::vector<std::thread> th;
        for (int l = 0; l < lanes; l++)
            th.emplace_back([&, l] {
                emul_wave = &w;
                const size_t i = base + l;
                u32 hw[8], rho[5], tau[5], neg;
                rd32(hw, h, i);
                fits[i] = sc_lattice_short(rho, tau, neg, hw) ? 1 : 0;
                tau_neg[i] = neg ? 1 : 0;
                memcpy(rho_out + 20 * i, rho, 20);
                memcpy(tau_out + 20 * i, tau, 20);
                emul_wave = nullptr;
            });
        for (auto& t : th) t.join();
        pthread_mutex_destroy(&w.mu);
        pthread_cond_destroy(&w.cv);
    }
}

// ed25519_Verify_Check over two wide combs (engine.hip: k_ed25519_verify_ctx_prepare / k_ed25519_verify_check_wide), one
// key for the batch: the key's comb built here with one addition per row, then per signature the kernel's lane code.
// Returns 1 if the path applies (key on the curve), 0 otherwise (verdicts untouched: the reference-order kernel's batch).
int emul_ed25519_verify_check_wide(int* verdict, const unsigned char* sig, const unsigned char* pk, const unsigned char* msg, size_t len,
                                   size_t n)
{
    u32 pkw[8], yw[8];
    rd32(pkw, pk, 0);
    for (int i = 0; i < 8; i++) yw[i] = pkw[i];
    const u32 parity = yw[7] >> 31;
    yw[7] &= 0x7fffffffu;
    ge_ext Q;
    fe_from_words(Q.Y, yw);
    if (!ge_calc_x_checked(Q.X, Q.Y, ~parity)) return 0;
    fe_mul(Q.T, Q.X, Q.Y);
    fe_set_u32(Q.Z, 1);
    ge_pe pe;
    ge_to_pe(pe, Q);                                                 // row 1 of the context: -A, Z = 1
    u32 row1[24];
    { u32 w[8]; fe_to_words(w, pe.ypx); memcpy(row1, w, 32); fe_to_words(w, pe.ymx); memcpy(row1 + 8, w, 32); fe_to_words(w, pe.t2d); memcpy(row1 + 16, w, 32); }
    ge_pa P;
    { u32 w[8]; memcpy(w, row1, 32); fe_from_words(P.ypx, w); memcpy(w, row1 + 8, 32); fe_from_words(P.ymx, w); memcpy(w, row1 + 16, 32); fe_from_words(P.t2d, w); }
    std::vector<u32> wide_key;
    build_wide_of(wide_key, P);
    for (size_t i = 0; i < n; i++) {
        u32 Rw[8], Sw[8], h[8], enc[8];
        rd32(Rw, sig, 2 * i);
        rd32(Sw, sig, 2 * i + 1);
        ed_hram(h, Rw, pkw, msg + len * i, len);
        sc_mod(h);
        unsigned short cs[WB_COLS], ch[WB_COLS];
        wb_columns(cs, 1, Sw);
        const u32 h_even = wb_columns<false>(ch, 1, h);
        ge_ext T;
        ge_double_base_mult_wide(T, wide_tables(), cs, wide_key.data(), ch, 1, h_even, row1);
        affine_pack_host(enc, T);
        u32 diff = 0;
        for (int j = 0; j < 8; j++) diff |= enc[j] ^ Rw[j];
        verdict[i] = diff == 0;
    }
    return 1;
}

// ed25519_Verify_Init: the 2080-byte context (pk || 16 canonical rows)
void emul_ed25519_verify_init(unsigned char* ctx, const unsigned char* pk, size_t n)
{
    for (size_t i = 0; i < n; i++) {
        u32 pkw[8];
        rd32(pkw, pk, i);
        memcpy(ctx + 2080 * i, pkw, 32);
        ge_ext Q;
        ed_decode_neg_key(Q, pkw);
        const QTableCanon t{ reinterpret_cast<u32*>(ctx + 2080 * i + 32) };
        qtable_build(t, Q);
    }
}

// ---- one operation per WAVE (csrc/coop_ops.cuh), every lane a fiber of this thread (coop_wave.h) ---------------------------
// The kernels of engine.hip are: LDS array, `if (blockIdx.x >= n) return`, the call below with e = blockIdx.x.
static std::mutex g_coop_mu;                 // one emulated workgroup at a time (the scheduler's stacks are shared)

unsigned long long emul_coop_sync_points(void) { return emul_coop::sync_points(); }

void emul_coop_x25519(unsigned char* out, const unsigned char* pk, unsigned char* sk, size_t n)
{
    std::lock_guard<std::mutex> lk(g_coop_mu);
    std::vector<u32> lds(coop::ROWQ_OFF);
    for (size_t e = 0; e < n; e++)
        emul_coop::run_block(64, [&] {
            const coop::Lane L = coop::make_lane(threadIdx.x);
            if (pk) coop::x25519_one<false>(lds.data(), L, out, pk, sk, e);
            else coop::x25519_one<true>(lds.data(), L, out, pk, sk, e);
        });
}

void emul_coop_x25519_two_waves(unsigned char* out, const unsigned char* pk, unsigned char* sk, size_t n)
{
    std::lock_guard<std::mutex> lk(g_coop_mu);
    std::vector<u32> lds(coop::X2_LDS_WORDS);
    for (size_t e = 0; e < n; e++)
        emul_coop::run_block(128, [&] { coop::x25519_two_waves(lds.data(), out, pk, sk, e); });
}

static void emul_fe_op_quad(unsigned char* out, const unsigned char* a, const unsigned char* b, size_t n, int op)
{
    std::lock_guard<std::mutex> lk(g_coop_mu);
    for (size_t base = 0; base < n; base += 16)
        emul_coop::run_block(64, [&] {
            const size_t i = base + (threadIdx.x >> 2);
            if (i >= n) return;
            u32 aw[8], bw[8], ow[8];
            rd32(aw, a, i); rd32(bw, b, i);
            fe_selftest_op(ow, aw, bw, op);
            if ((threadIdx.x & 3) == 0) wr32(out, i, ow);
        });
}

// ---- four lanes per element (csrc/quad25519.cuh): a wave of 64 lock-step lanes carries 16 elements; the kernel is
// `e = blockIdx.x * 16 + lane / 4; if (e >= n) return;` around the call below
void emul_quad_x25519(unsigned char* out, const unsigned char* pk, unsigned char* sk, size_t n)
{
    std::lock_guard<std::mutex> lk(g_coop_mu);
    for (size_t base = 0; base < n; base += quad::ELEMS_PER_WAVE)
        emul_coop::run_block(64, [&] {
            const size_t e = base + (threadIdx.x >> 2);
            if (e >= n) return;
            if (pk) quad::x25519_element<false>(out, pk, sk, e);
            else quad::x25519_element<true>(out, pk, sk, e);
        });
}

// the fixed-base operations on quads (k_ed25519_keypair_quad / k_ed25519_sign_quad / k_x25519_public_fast_quad): the kernels are the
// element index, `if (e >= n) return`, the lanes' parked columns in LDS, and the call below
void emul_quad_keypair(unsigned char* pub, unsigned char* priv, const unsigned char* sk, size_t n)
{
    std::lock_guard<std::mutex> lk(g_coop_mu);
    const u32* wide = wide_tables();
    std::vector<unsigned short> cols(WB_COLS * 64);
    for (size_t base = 0; base < n; base += quad::ELEMS_PER_WAVE)
        emul_coop::run_block(64, [&] {
            const size_t e = base + (threadIdx.x >> 2);
            if (e >= n) return;
            quad::keypair_element(pub, priv, sk, e, wide, cols.data() + threadIdx.x, 64);
        });
}

void emul_quad_sign(unsigned char* sig, const unsigned char* priv, const unsigned char* msg, size_t len, size_t n)
{
    std::lock_guard<std::mutex> lk(g_coop_mu);
    const u32* wide = wide_tables();
    std::vector<unsigned short> cols(WB_COLS * 64);
    for (size_t base = 0; base < n; base += quad::ELEMS_PER_WAVE)
        emul_coop::run_block(64, [&] {
            const size_t e = base + (threadIdx.x >> 2);
            if (e >= n) return;
            quad::sign_element(sig, priv, msg + len * e, len, e, wide, cols.data() + threadIdx.x, 64);
        });
}

void emul_quad_public_fast(unsigned char* pk, unsigned char* sk, size_t n)
{
    std::lock_guard<std::mutex> lk(g_coop_mu);
    const u32* wide = wide_tables();
    std::vector<unsigned short> cols(WB_COLS * 64);
    for (size_t base = 0; base < n; base += quad::ELEMS_PER_WAVE)
        emul_coop::run_block(64, [&] {
            const size_t e = base + (threadIdx.x >> 2);
            if (e >= n) return;
            quad::public_fast_element(pk, sk, e, wide, cols.data() + threadIdx.x, 64);
        });
}

// ed25519_Verify_Check over two wide combs on quads (k_ed25519_verify_check_wide_quad: quad::verify_check_wide_element): the key's
// context by the emulated Verify_Init, its comb as emul_ed25519_verify_check_wide builds it.  Returns 1 if the path applies.
int emul_quad_verify_check_wide(int* verdict, const unsigned char* sig, const unsigned char* pk, const unsigned char* msg, size_t len, size_t n)
{
    u32 pkw[8], yw[8];
    rd32(pkw, pk, 0);
    for (int i = 0; i < 8; i++) yw[i] = pkw[i];
    const u32 parity = yw[7] >> 31;
    yw[7] &= 0x7fffffffu;
    ge_ext Q;
    fe_from_words(Q.Y, yw);
    if (!ge_calc_x_checked(Q.X, Q.Y, ~parity)) return 0;
    std::vector<u32> ctx(2080 / 4);
    emul_ed25519_verify_init(reinterpret_cast<unsigned char*>(ctx.data()), pk, 1);
    ge_pa P;
    { u32 w[8]; memcpy(w, ctx.data() + 8 + 32, 32); fe_from_words(P.ypx, w); memcpy(w, ctx.data() + 8 + 40, 32); fe_from_words(P.ymx, w);
      memcpy(w, ctx.data() + 8 + 48, 32); fe_from_words(P.t2d, w); }
    std::vector<u32> wide_key;
    build_wide_of(wide_key, P);
    std::lock_guard<std::mutex> lk(g_coop_mu);
    const u32* wide = wide_tables();
    std::vector<unsigned short> cols(2 * WB_COLS * 64);
    for (size_t base = 0; base < n; base += quad::ELEMS_PER_WAVE)
        emul_coop::run_block(64, [&] {
            const size_t e = base + (threadIdx.x >> 2);
            if (e >= n) return;
            quad::verify_check_wide_element(verdict, sig, ctx.data(), msg + len * e, len, e, wide, wide_key.data(), cols.data() + threadIdx.x,
                                            cols.data() + WB_COLS * 64 + threadIdx.x, 64);
        });
    return 1;
}

// the lattice path with the WALK on quads (quad::walk_is_neutral): scalars, decoding and window tables by the one-lane code, as
// k_ed25519_verify_quad_prep runs them (side by side: the points are tabulated as decoded, tau's sign reaches the walk as a flip of
// the key rows' signs); then 16 elements per wave walk together from the wave's top digit (k_ed25519_verify_quad_walk).  need_slow: the elements the walk does not decide (their verdict stays 0).
void emul_quad_verify(int* verdict, int* need_slow, const unsigned char* sig, const unsigned char* pk, const unsigned char* msg,
                      size_t len, size_t n)
{
    std::lock_guard<std::mutex> lk(g_coop_mu);
    const u32* tbl = tables() + (size_t)SC_TBL_OFFSET;
    constexpr int G = quad::ELEMS_PER_WAVE;
    std::vector<u32> tabs((size_t)G * 2 * WTABLE_WORDS), cols((size_t)SIGMA_WORDS * G), rho(5 * G), tau(5 * G);
    for (size_t base = 0; base < n; base += G) {
        const int m = (int)std::min<size_t>(G, n - base);
        int wave_top = 0;
        u32 r_ok[G] = {}, walks[G] = {}, flip[G] = {};
        for (int j = 0; j < m; j++) {
            const size_t i = base + j;
            u32 pkw[8], Rw[8], Sw[8], c[SIGMA_WORDS], rh[5], ta[5], tau_neg;
            rd32(pkw, pk, i);
            rd32(Rw, sig, 2 * i);
            rd32(Sw, sig, 2 * i + 1);
            const u32 lat_ok = ed_verify_fast_scalars(c, rh, ta, tau_neg, pkw, Rw, Sw, msg + len * i, len);
            fe QX, QY, RX, RY;
            const u32 q_ok = ed_verify_fast_decode(QX, QY, pkw, 0u, 0u);       // as k_ed25519_verify_quad_prep: tau's sign is not known yet
            r_ok[j] = ed_verify_fast_decode(RX, RY, Rw, 0xffffffffu, 0u);
            flip[j] = tau_neg;                                                 // ... the walk flips the key rows' signs instead
            need_slow[i] = (lat_ok && q_ok) ? 0 : 1;
            verdict[i] = 0;
            walks[j] = !need_slow[i];
            if (!walks[j]) continue;
            wtable_build(tabs.data() + (size_t)j * 2 * WTABLE_WORDS, QX, QY);
            wtable_build(tabs.data() + (size_t)j * 2 * WTABLE_WORDS + WTABLE_WORDS, RX, RY);
            for (int w = 0; w < SIGMA_WORDS; w++) cols[(size_t)w * G + j] = c[w];
            for (int w = 0; w < 5; w++) { rho[(size_t)w * G + j] = rh[w]; tau[(size_t)w * G + j] = ta[w]; }
            wave_top = std::max(wave_top, walk_top_digit(ta, rh));
        }
        if (wave_top < 8) wave_top = 8;
        emul_coop::run_block(64, [&] {
            const int j = (int)(threadIdx.x >> 2);
            if (j >= m || !walks[j]) return;
            const quad::Roles R = quad::roles();
            const WalkScalars sc{ cols.data(), tau.data(), rho.data(), (size_t)G, (size_t)j };
            const u32* tq = tabs.data() + (size_t)j * 2 * WTABLE_WORDS;
            const u32 neutral = quad::walk_is_neutral(sc, tq, tq + WTABLE_WORDS, tbl, wave_top, R, flip[j]);
            if (R.is0) verdict[base + j] = (r_ok[j] && neutral) ? 1 : 0;
        });
    }
}

void emul_coop_public_fast(unsigned char* pk, unsigned char* sk, size_t n, int wide)
{
    std::lock_guard<std::mutex> lk(g_coop_mu);
    std::vector<u32> lds(coop::LDS_WORDS);
    const u32* tbl = wide ? wide_tables() : tables();
    for (size_t e = 0; e < n; e++)
        emul_coop::run_block(64, [&] {
            const coop::Lane L = coop::make_lane(threadIdx.x);
            if (wide) coop::public_fast_one<true>(lds.data(), L, pk, sk, e, tbl);
            else coop::public_fast_one<false>(lds.data(), L, pk, sk, e, tbl);
        });
}

void emul_coop_keypair(unsigned char* pub, unsigned char* priv, const unsigned char* blinding, const unsigned char* sk, size_t n, int wide)
{
    std::lock_guard<std::mutex> lk(g_coop_mu);
    std::vector<u32> lds(coop::LDS_WORDS);
    const u32* tbl = wide ? wide_tables() : tables();
    for (size_t e = 0; e < n; e++)
        emul_coop::run_block(64, [&] {
            const coop::Lane L = coop::make_lane(threadIdx.x);
            if (wide) coop::keypair_one<true>(lds.data(), L, pub, priv, sk, e, tbl, reinterpret_cast<const u32*>(blinding));
            else coop::keypair_one<false>(lds.data(), L, pub, priv, sk, e, tbl, nullptr);
        });
}

void emul_coop_sign(unsigned char* sig, const unsigned char* priv, const unsigned char* blinding, const unsigned char* msg, size_t len,
                    size_t n, int wide)
{
    std::lock_guard<std::mutex> lk(g_coop_mu);
    std::vector<u32> lds(coop::LDS_WORDS);
    const u32* tbl = wide ? wide_tables() : tables();
    const Msgs msgs{ msg, len, nullptr };
    std::vector<u64> sha_wk(80);                         // as k_ed25519_sign_coop: the second wave serves every compression of the three hashes
    for (size_t e = 0; e < n; e++)
        emul_coop::run_block(128, [&] {
            if (threadIdx.x >= 64) { coop::sha_schedule_server(sha_wk.data(), 1 + sha512_blocks(4, len) + sha512_blocks(8, len)); return; }
            const coop::Lane L = coop::make_lane(threadIdx.x);
            const coop::ShaTwoWaves sha{ sha_wk.data() };
            if (wide) coop::sign_one<true>(lds.data(), L, sig, priv, msgs, e, tbl, reinterpret_cast<const u32*>(blinding), nullptr, sha);
            else coop::sign_one<false>(lds.data(), L, sig, priv, msgs, e, tbl, nullptr, nullptr, sha);
        });
}

void emul_coop_blinding_init(unsigned char* ctx /* 192 */, const unsigned char* seed, size_t len)
{
    std::lock_guard<std::mutex> lk(g_coop_mu);
    std::vector<u32> lds(coop::LDS_WORDS);
    emul_coop::run_block(64, [&] {
        coop::blinding_init_one(lds.data(), coop::make_lane(threadIdx.x), reinterpret_cast<u32*>(ctx), seed, len, wide_tables());
    });
}

// n contexts of 2080 bytes (pk || 16 rows), the rows by the wave
void emul_coop_verify_init(unsigned char* ctx, const unsigned char* pk, size_t n)
{
    std::lock_guard<std::mutex> lk(g_coop_mu);
    std::vector<u32> lds(coop::Q_LDS_WORDS);
    for (size_t e = 0; e < n; e++) {
        memcpy(ctx + 2080 * e, pk + 32 * e, 32);
        emul_coop::run_block(64, [&] {
            coop::verify_init_one(lds.data(), coop::make_lane(threadIdx.x), pk, e, reinterpret_cast<u32*>(ctx + 2080 * e + 32));
        });
    }
}

// one context, n (signature, message) pairs
void emul_coop_verify_check(int* verdict, const unsigned char* ctx, const unsigned char* sig, const unsigned char* msg, size_t len, size_t n)
{
    std::lock_guard<std::mutex> lk(g_coop_mu);
    std::vector<u32> lds(coop::Q_LDS_WORDS);
    const Msgs msgs{ msg, len, nullptr };
    for (size_t e = 0; e < n; e++)
        emul_coop::run_block(64, [&] {
            coop::verify_check_one(lds.data(), coop::make_lane(threadIdx.x), verdict, sig, reinterpret_cast<const u32*>(ctx), msgs, e,
                                   tables() + (size_t)REF_TBL_OFFSET);
        });
}

// the lattice path, three waves per element; need_slow[e] = 1 where the element went on the slow list (verdict untouched)
void emul_coop_verify_three_waves(int* verdict, int* need_slow, const unsigned char* sig, const unsigned char* pk, const unsigned char* msg,
                                  size_t len, size_t n, int lat_cap_bits)
{
    std::lock_guard<std::mutex> lk(g_coop_mu);
    std::vector<u32> lds(coop::V3_LDS_WORDS), park(40), hand(4);
    std::vector<u32> sigma((size_t)SIGMA_WORDS * n), rho(5 * n), tau(5 * n), flags(n), slow_list(n), counters(4, 0);
    FastScratch fs{};
    fs.sigma = sigma.data(); fs.rho = rho.data(); fs.tau = tau.data(); fs.flags = flags.data();
    fs.slow_list = slow_list.data(); fs.slow_count = counters.data();
    fs.lat_cap_bits = lat_cap_bits > 0 ? lat_cap_bits : LAT_CAP_BITS;
    const Msgs msgs{ msg, len, nullptr };
    for (size_t e = 0; e < n; e++)
        emul_coop::run_block(192, [&] {
            coop::verify_three_waves(lds.data(), park.data(), hand.data(), fs, verdict, sig, pk, msgs, n, e, tables());
        });
    for (size_t e = 0; e < n; e++) need_slow[e] = (flags[e] & FLAG_SLOW) ? 1 : 0;
}

}  // extern "C"
