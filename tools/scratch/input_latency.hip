// tools/scratch/input_latency.hip -- what 64 input bytes cost a one-wave kernel: read from pinned host memory by the kernel's first
// loads (what a zero-copy call does) against passed by value in the kernel arguments.  Both end with a completion word the host spins on.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
struct In64 { uint4 v[4]; };
__global__ void k_ptr(uint32_t* out, const uint4* in, volatile uint32_t* flag, uint32_t seq)
{
    uint4 a = in[0], b = in[1], c = in[2], d = in[3];
    if (threadIdx.x == 0) { out[0] = a.x ^ b.y ^ c.z ^ d.w; __threadfence_system(); *flag = seq; }
}
__global__ void k_val(uint32_t* out, In64 in, volatile uint32_t* flag, uint32_t seq)
{
    if (threadIdx.x == 0) { out[0] = in.v[0].x ^ in.v[1].y ^ in.v[2].z ^ in.v[3].w; __threadfence_system(); *flag = seq; }
}
int main()
{
    uint32_t *out, *flag; uint4* in;
    CK(hipHostMalloc(&out, 64, hipHostMallocDefault));
    CK(hipHostMalloc(&flag, 64, hipHostMallocDefault));
    CK(hipHostMalloc(&in, 64, hipHostMallocDefault));
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    auto now = [] { return std::chrono::steady_clock::now(); };
    const int reps = 2000;
    double t[2] = { 0, 0 };
    for (int mode = 0; mode < 2; mode++)
        for (int r = -100; r < reps; r++) {
            in[0].x = r; in[1].y = 2 * r; in[2].z = 3 * r; in[3].w = 5 * r;
            In64 v; for (int i = 0; i < 4; i++) v.v[i] = in[i];
            auto a = now();
            if (mode == 0) k_ptr<<<1, 64, 0, st>>>(out, in, flag, (uint32_t)(r + 1000));
            else k_val<<<1, 64, 0, st>>>(out, v, flag, (uint32_t)(r + 1000));
            while (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != (uint32_t)(r + 1000)) { }
            auto b = now();
            if (out[0] != (uint32_t)(r ^ (2 * r) ^ (3 * r) ^ (5 * r))) { printf("wrong result\n"); return 1; }
            if (r >= 0) t[mode] += std::chrono::duration<double, std::micro>(b - a).count();
        }
    printf("64 input bytes read from pinned host memory: %.2f us per call | passed by value in the kernel arguments: %.2f us per call\n", t[0] / reps, t[1] / reps);
    return 0;
}
