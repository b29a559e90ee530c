// tools/scratch/salu_rate.hip -- issue cost of scalar ALU instructions in a lone wave (one wave per SIMD), in shader cycles
// per instruction: one dependent chain, three interleaved chains, and a VALU chain for comparison.
#include <hip/hip_runtime.h>
#include <cstdio>
#define R8(x) x x x x x x x x
__global__ void __launch_bounds__(64) k(unsigned long long* out, int iters)
{
    unsigned a = threadIdx.x >> 8, b = 3, c = 5, d = 7;     // uniform
    a = __builtin_amdgcn_readfirstlane(a); 
    unsigned long long t0, t1, t2, t3;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0));
    for (int i = 0; i < iters; i++)
        asm volatile(R8(R8("s_add_u32 %0, %0, %1\n\ts_xor_b32 %0, %0, %2\n\t")) : "+s"(a) : "s"(b), "s"(c) : "scc");          // 128 dependent
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1));
    for (int i = 0; i < iters; i++)
        asm volatile(R8(R8("s_add_u32 %0, %0, %3\n\ts_xor_b32 %1, %1, %3\n\ts_sub_u32 %2, %2, %3\n\t")) : "+s"(a), "+s"(b), "+s"(c) : "s"(d) : "scc");   // 192, three chains
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t2));
    unsigned v = threadIdx.x, w = 9;
    for (int i = 0; i < iters; i++)
        asm volatile(R8(R8("v_add_u32 %0, %0, %1\n\tv_xor_b32 %0, %0, %1\n\t")) : "+v"(v) : "v"(w));                    // 128 dependent VALU
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t3));
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = t2 - t1; out[2] = t3 - t2; out[3] = a + b + c; }
    if (v == 0x12345) out[4] = v;
}
int main()
{
    unsigned long long* d; hipMalloc(&d, 64);
    const int iters = 200;
    for (int waves = 1; waves <= 4; waves *= 2) {
        // `waves` single-wave workgroups per SIMD: 256 CUs x 4 SIMDs x waves
        k<<<256 * 4 * waves, 64>>>(d, iters); hipDeviceSynchronize();
        k<<<256 * 4 * waves, 64>>>(d, iters); hipDeviceSynchronize();
        unsigned long long h[5]; hipMemcpy(h, d, 40, hipMemcpyDeviceToHost);
        printf("%d wave(s) per SIMD: SALU one chain %.2f cycles/instr, SALU three chains %.2f, VALU one chain %.2f\n", waves,
               (double)h[0] / (iters * 128), (double)h[1] / (iters * 192), (double)h[2] / (iters * 128));
    }
    return 0;
}
