// tools/scratch/coop_stamps.hip -- where the cycles of ONE ed25519_SignMessage / ed25519_CreateKeyPair go inside the per-wave kernel
// (coop_ops.cuh: sign_one / keypair_one, copied here with s_memtime stamps between their parts).  Timing experiment only.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -pragma-unroll-threshold=131072 -I curve25519_amd/csrc tools/scratch/coop_stamps.hip -o tools/scratch/coop_stamps
#include "engine_common.cuh"
#include <cstdio>
#include <vector>
using namespace c25519::coop;

__device__ unsigned long long g_stamps[16];
#define STAMP(i) do { if (threadIdx.x == 0) g_stamps[i] = __builtin_amdgcn_s_memrealtime(); __builtin_amdgcn_s_waitcnt(0); } while (0)

extern "C" __global__ void __launch_bounds__(128) k_sign_stamped(void* sig, const void* priv, const uint8_t* msg, size_t len, const u32* wide)
{
    __shared__ __attribute__((aligned(16))) u32 lds[LDS_WORDS];
    __shared__ u64 sha_wk[80];
    if (threadIdx.x >= 64) { sha_schedule_server(sha_wk, 1 + sha512_blocks(4, len) + sha512_blocks(8, len)); return; }
    const ShaTwoWaves sha{ sha_wk };
    const Lane L = make_lane(threadIdx.x);
    STAMP(0);
    u32 seed[8], pkw[8], a[8], r[8], xw[8], yw[8], enc[8], s[8];
    load32(seed, priv, 0);
    load32(pkw, priv, 1);
    STAMP(1);
    u64 b_words[4], dg[8];
    u32 le[16];
    ed_expand_seed(a, b_words, seed, sha);
    STAMP(2);
    sha512_prefixed<4>(dg, b_words, msg, len, sha);
    STAMP(3);
    sha512_digest_le_words(le, dg);
    sc_reduce512(r, le);
    sc_mod(r);
    STAMP(4);
    setup_one(lds, L);
    STAMP(5);
    const u32 v = ge_base_mult_wide(lds, L, r, wide, nullptr);
    STAMP(6);
    ge_affine_words(xw, yw, lds, L, v);
    STAMP(7);
    ge_pack(enc, xw, yw);
    ed_sign_s(s, enc, pkw, msg, len, a, r, sha);
    STAMP(8);
    if (threadIdx.x == 0) { store32(sig, 0, enc); store32(sig, 1, s); }
    wipe(lds, LDS_WORDS);
    STAMP(9);
}

int main()
{
    const u32* wide = nullptr;
    if (wide_tables(&wide)) { printf("no tables: %s\n", c25519_amd_last_error()); return 1; }
    unsigned char *priv, *msg, *sig;
    hipMalloc(&priv, 64); hipMalloc(&msg, 32); hipMalloc(&sig, 64);
    unsigned char sk[32], pub[32], pr[64], m[32];
    for (int i = 0; i < 32; i++) { sk[i] = (unsigned char)(3 * i + 1); m[i] = (unsigned char)(7 * i); }
    ed25519_CreateKeyPair(pub, pr, nullptr, sk);
    hipMemcpy(priv, pr, 64, hipMemcpyHostToDevice); hipMemcpy(msg, m, 32, hipMemcpyHostToDevice);
    unsigned long long st[16], best[16] = {};
    for (int rep = 0; rep < 20; rep++) {
        k_sign_stamped<<<1, 128>>>(sig, priv, msg, 32, wide);
        hipDeviceSynchronize();
        hipMemcpyFromSymbol(st, HIP_SYMBOL(g_stamps), sizeof st);
        for (int i = 1; i < 10; i++) { const unsigned long long d = st[i] - st[i - 1]; if (rep == 0 || d < best[i]) best[i] = d; }
    }
    unsigned char ref[64], got[64];
    ed25519_SignMessage(ref, pr, nullptr, m, 32);
    hipMemcpy(got, sig, 64, hipMemcpyDeviceToHost);
    const char* names[] = { "", "load priv", "SHA-512 block A (seed) + clamp", "SHA-512 block B (nonce)", "r mod L", "setup_one", "walk over the wide comb (columns, row fetch, 47 levels)",
                            "affine: inversion + 1 level + canonical words", "pack, SHA-512 block C, S = h a + r mod L", "store, LDS wipe" };
    unsigned long long tot = 0;
    for (int i = 1; i < 10; i++) tot += best[i];
    printf("k_sign_stamped: %s; s_memrealtime ticks (100 MHz: 10 ns), best of 20\n", memcmp(ref, got, 64) ? "WRONG BYTES" : "bytes = the library's");
    for (int i = 1; i < 10; i++) printf("  %-62s %7.2f us\n", names[i], best[i] * 0.01);
    printf("  %-62s %7.2f us\n", "total", tot * 0.01);
    return 0;
}
