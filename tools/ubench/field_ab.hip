// tools/ubench/field_ab.hip -- the same X25519 ladder step (5 M + 4 S + a24 step + 8 add/sub + the per-bit select,
// curve25519_dh.c:57-84) timed in three field representations on the device it runs on:
//   A  10 x 25.5-bit limbs, chained v_mad_u64_u32 columns         (the product: curve25519_amd/csrc/fe25519.cuh)
//   B  8 x 32-bit saturated words, v_mad_u64_u32 + v_addc_co_u32  (north_star's / the reference's shape)
//   C  9 x 28.33-bit limbs, 17 columns + double fold              (fewest products)
// One lane per ladder, 2^20 lanes, STEPS ladder steps each (a full X25519 is 255).  Prints ms per 2^20 x 255 steps --
// directly comparable with k_x25519_fused's time per pass -- and the implied ladders per second.  B and C are
// validated on the CPU against big integers and the RFC 7748 vectors (tests/test_field_forms.py).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I curve25519_amd/csrc tools/ubench/field_ab.hip -o tools/ubench/field_ab
#include "field_forms.cuh"
#include "x25519.cuh"

#include <cstdio>
#include <cstdlib>

using namespace c25519;

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
    fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

constexpr int STEPS = 255;

C25519_DEV void load_words(u32 (&w)[8], const u32* p, size_t i)
{
#pragma unroll
    for (int j = 0; j < 8; j++) w[j] = p[8 * i + j];
}

// FORM 0: product field; 8: fe8; 9: fe9.  The scalar bits come from k so that the select is a real data dependence.
template <int FORM>
__global__ void __launch_bounds__(256, 4) k_steps(u32* out, const u32* u, const u32* k, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    u32 uw[8], kw[8], ow[8];
    load_words(uw, u, i);
    load_words(kw, k, i);
    u32 prev = 1;
    if constexpr (FORM == 0) {
        fe X1, SX, SZ, DX, DZ;
        fe_from_words(X1, uw);
        SX = X1; fe_set_u32(SZ, 1); DX = X1; fe_set_u32(DZ, 1);
        mont_double(DX, DZ);
#pragma unroll 1
        for (int b = STEPS - 1; b >= 0; b--) {
            const u32 bit = (kw[(b >> 5) & 7] >> (b & 31)) & 1u;
            ladder_step<false>(SX, SZ, DX, DZ, X1, (u32)0 - (u32)(bit == prev));
            prev = bit;
        }
        fe_mul(SX, SX, DZ); fe_mul(SZ, SZ, DX); fe_add(SX, SX, SZ);
        fe_to_words(ow, SX);
    } else if constexpr (FORM == 8) {
        fe8 X1, SX, SZ, DX, DZ;
        const u32 one[8] = { 1, 0, 0, 0, 0, 0, 0, 0 };
        fe8_from_words(X1, uw);
        SX = X1; fe8_from_words(SZ, one); DX = X1; fe8_from_words(DZ, one);
#pragma unroll 1
        for (int b = STEPS - 1; b >= 0; b--) {
            const u32 bit = (kw[(b >> 5) & 7] >> (b & 31)) & 1u;
            ladder_step8(SX, SZ, DX, DZ, X1, (u32)0 - (u32)(bit == prev));
            prev = bit;
        }
        fe8_mul(SX, SX, DZ); fe8_mul(SZ, SZ, DX); fe8_add(SX, SX, SZ);
        fe8_to_words(ow, SX);
    } else {
        fe9 X1, SX, SZ, DX, DZ;
        const u32 one[8] = { 1, 0, 0, 0, 0, 0, 0, 0 };
        fe9_from_words(X1, uw);
        SX = X1; fe9_from_words(SZ, one); DX = X1; fe9_from_words(DZ, one);
#pragma unroll 1
        for (int b = STEPS - 1; b >= 0; b--) {
            const u32 bit = (kw[(b >> 5) & 7] >> (b & 31)) & 1u;
            ladder_step9(SX, SZ, DX, DZ, X1, (u32)0 - (u32)(bit == prev));
            prev = bit;
        }
        fe9 t;
        fe9_mul(SX, SX, DZ); fe9_mul(SZ, SZ, DX); fe9_add(t, SX, SZ);
        fe9_to_words(ow, t);
    }
#pragma unroll
    for (int j = 0; j < 8; j++) out[8 * i + j] = ow[j];
}

// the three multiplications / squarings alone (MULS back-to-back dependent operations per lane)
template <int FORM, bool SQR>
__global__ void __launch_bounds__(256, 4) k_mulsqr(u32* out, const u32* u, const u32* k, size_t n, int reps)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    u32 uw[8], kw[8], ow[8];
    load_words(uw, u, i);
    load_words(kw, k, i);
    if constexpr (FORM == 0) {
        fe a, b; fe_from_words(a, uw); fe_from_words(b, kw);
#pragma unroll 1
        for (int r = 0; r < reps; r++) { if (SQR) fe_sqr(a, a); else fe_mul(a, a, b); }
        fe_to_words(ow, a);
    } else if constexpr (FORM == 8) {
        fe8 a, b; fe8_from_words(a, uw); fe8_from_words(b, kw);
#pragma unroll 1
        for (int r = 0; r < reps; r++) { if (SQR) fe8_sqr(a, a); else fe8_mul(a, a, b); }
        fe8_to_words(ow, a);
    } else {
        fe9 a, b; fe9_from_words(a, uw); fe9_from_words(b, kw);
#pragma unroll 1
        for (int r = 0; r < reps; r++) { if (SQR) fe9_sqr(a, a); else fe9_mul(a, a, b); }
        fe9_to_words(ow, a);
    }
#pragma unroll
    for (int j = 0; j < 8; j++) out[8 * i + j] = ow[j];
}

template <typename F>
static float time_it(F launch, hipStream_t s)
{
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    launch(); CHECK(hipStreamSynchronize(s));
    float best = 1e30f, sum = 0;
    const int R = 5;
    for (int r = 0; r < R; r++) {
        CHECK(hipEventRecord(e0, s)); launch(); CHECK(hipEventRecord(e1, s)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best; sum += ms;
    }
    (void)sum;
    return best;
}

int main()
{
    const size_t n = (size_t)1 << 20;
    u32 *u, *k, *o;
    CHECK(hipMalloc(&u, 32 * n)); CHECK(hipMalloc(&k, 32 * n)); CHECK(hipMalloc(&o, 32 * n));
    u32* h = (u32*)malloc(32 * n);
    unsigned long long x = 0x9E3779B97F4A7C15ull;
    for (size_t i = 0; i < 8 * n; i++) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; h[i] = (u32)(x >> 16); }
    CHECK(hipMemcpy(u, h, 32 * n, hipMemcpyHostToDevice));
    for (size_t i = 0; i < 8 * n; i++) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; h[i] = (u32)(x >> 16); }
    CHECK(hipMemcpy(k, h, 32 * n, hipMemcpyHostToDevice));
    hipStream_t s; CHECK(hipStreamCreate(&s));
    const unsigned grid = (unsigned)(n / 256);
    printf("X25519 ladder step (5M + 4S + a24 + 8 add/sub + select) per lane, 2^20 lanes x %d steps, best of 5\n", STEPS);
    printf("%-52s %10s %14s\n", "field representation", "ms/pass", "M ladders/s");
    struct { const char* name; float ms; } rows[3];
    rows[0] = { "A 10 x 25.5-bit, chained MAD columns (product)", time_it([&] { k_steps<0><<<grid, 256, 0, s>>>(o, u, k, n); }, s) };
    rows[1] = { "B  8 x 32-bit saturated, MAD + addc", time_it([&] { k_steps<8><<<grid, 256, 0, s>>>(o, u, k, n); }, s) };
    rows[2] = { "C  9 x 28.33-bit, 17 columns + double fold", time_it([&] { k_steps<9><<<grid, 256, 0, s>>>(o, u, k, n); }, s) };
    for (auto& r : rows) printf("%-52s %10.3f %14.1f\n", r.name, r.ms, n / (r.ms * 1e-3) / 1e6);
    const int reps = 2048;
    printf("\n%d dependent multiplications / squarings per lane, 2^20 lanes: ns per operation per 2^20 lanes\n", reps);
    printf("%-52s %12s %12s\n", "field representation", "mul", "sqr");
    float m0 = time_it([&] { k_mulsqr<0, false><<<grid, 256, 0, s>>>(o, u, k, n, reps); }, s);
    float s0 = time_it([&] { k_mulsqr<0, true><<<grid, 256, 0, s>>>(o, u, k, n, reps); }, s);
    float m8 = time_it([&] { k_mulsqr<8, false><<<grid, 256, 0, s>>>(o, u, k, n, reps); }, s);
    float s8 = time_it([&] { k_mulsqr<8, true><<<grid, 256, 0, s>>>(o, u, k, n, reps); }, s);
    float m9 = time_it([&] { k_mulsqr<9, false><<<grid, 256, 0, s>>>(o, u, k, n, reps); }, s);
    float s9 = time_it([&] { k_mulsqr<9, true><<<grid, 256, 0, s>>>(o, u, k, n, reps); }, s);
    printf("%-52s %12.1f %12.1f\n", "A 10 x 25.5-bit", m0 * 1e6 / reps, s0 * 1e6 / reps);
    printf("%-52s %12.1f %12.1f\n", "B  8 x 32-bit saturated", m8 * 1e6 / reps, s8 * 1e6 / reps);
    printf("%-52s %12.1f %12.1f\n", "C  9 x 28.33-bit", m9 * 1e6 / reps, s9 * 1e6 / reps);
    return 0;
}
