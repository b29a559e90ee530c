// scratch: what v_permlane16_swap / v_permlane32_swap / DPP row_shr / row_ror do, lane by lane
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out)
{
    unsigned x = threadIdx.x, y = threadIdx.x + 100;
    auto r = __builtin_amdgcn_permlane16_swap(x, y, false, false);
    auto r2 = __builtin_amdgcn_permlane32_swap(x, y, false, false);
    out[threadIdx.x] = r[0]; out[64 + threadIdx.x] = r[1];
    out[128 + threadIdx.x] = r2[0]; out[192 + threadIdx.x] = r2[1];
    out[256 + threadIdx.x] = __builtin_amdgcn_update_dpp(999u, x, 0x111, 0xf, 0xf, true);   // row_shr:1
    out[320 + threadIdx.x] = __builtin_amdgcn_update_dpp(999u, x, 0x127, 0xf, 0xf, true);   // row_ror:7
    out[384 + threadIdx.x] = __builtin_amdgcn_update_dpp(999u, x, 0x112, 0xf, 0xf, true);   // row_shr:2
    out[448 + threadIdx.x] = __builtin_amdgcn_update_dpp(999u, x, 0x128, 0xf, 0xf, true);   // row_ror:8
}
int main()
{
    unsigned* d; hipMalloc(&d, 512 * 4);
    k<<<1, 64>>>(d);
    unsigned h[512]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    const char* names[] = { "permlane16_swap[0] (x)", "permlane16_swap[1] (y)", "permlane32_swap[0]", "permlane32_swap[1]", "row_shr:1", "row_ror:7", "row_shr:2", "row_ror:8" };
    for (int q = 0; q < 8; q++) { printf("%-24s", names[q]); for (int i = 0; i < 64; i++) printf(" %u", h[q * 64 + i]); printf("\n"); }
    return 0;
}
