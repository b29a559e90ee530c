// tools/scratch/launch_latency.hip -- what the host pays around ONE tiny kernel: launch + event record + event synchronise (what a
// single call through the reference's prototypes does today) against launch + a completion word in pinned host memory that the
// kernel's last store sets and the host spins on.   hipcc --offload-arch=gfx950 -O3 launch_latency.hip -o launch_latency
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k_work(uint32_t* out, volatile uint32_t* flag, uint32_t seq, int spin)
{
    uint32_t x = threadIdx.x;
    for (int i = 0; i < spin; i++) x = x * 1664525u + 1013904223u;      // a dependent chain: ~8 cycles a trip
    if (threadIdx.x == 0) {
        out[0] = x;
        if (flag) { __threadfence_system(); *flag = seq; }
    }
}

int main()
{
    uint32_t *out, *flag;
    CK(hipHostMalloc(&out, 64, hipHostMallocDefault));
    CK(hipHostMalloc(&flag, 64, hipHostMallocDefault));
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipEvent_t ev; CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto us = [](auto a, auto b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
    for (int spin : { 0, 10000, 30000 }) {
        const int reps = 400;
        double t_ev = 0, t_ss = 0, t_flag = 0, t_wv = 0;
        for (int mode = 0; mode < 4; mode++) {
            for (int r = -50; r < reps; r++) {
                *flag = 0;
                auto a = now();
                if (mode == 0) {
                    k_work<<<1, 64, 0, st>>>(out, nullptr, 0, spin);
                    CK(hipEventRecord(ev, st));
                    CK(hipEventSynchronize(ev));
                } else if (mode == 1) {
                    k_work<<<1, 64, 0, st>>>(out, nullptr, 0, spin);
                    CK(hipStreamSynchronize(st));
                } else if (mode == 3) {
                    k_work<<<1, 64, 0, st>>>(out, nullptr, 0, spin);
                    CK(hipStreamWriteValue32(st, flag, (uint32_t)(r + 100), 0));
                    while (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != (uint32_t)(r + 100)) { }
                } else {
                    k_work<<<1, 64, 0, st>>>(out, flag, (uint32_t)(r + 100), spin);
                    while (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != (uint32_t)(r + 100)) { }
                }
                auto b = now();
                if (r >= 0) (mode == 0 ? t_ev : mode == 1 ? t_ss : mode == 3 ? t_wv : t_flag) += us(a, b);
            }
            CK(hipStreamSynchronize(st));
        }
        printf("kernel of %5d trips: launch + event record + event sync %7.2f us | launch + stream sync %7.2f us | launch + spin on a pinned word %7.2f us | launch + hipStreamWriteValue32 + spin %7.2f us\n",
               spin, t_ev / reps, t_ss / reps, t_flag / reps, t_wv / reps);
    }
    return 0;
}
