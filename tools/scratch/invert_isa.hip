// tools/scratch/invert_isa.hip -- the two inversions side by side for an instruction count (hipcc -S) and a timing
#include "fe25519.cuh"
using namespace c25519;
__global__ void __launch_bounds__(64) k_inv_fermat(u32* out, const u32* in, int reps)
{
    u32 w[8];
    for (int i = 0; i < 8; i++) w[i] = in[threadIdx.x * 8 + i];
    fe z, r;
    fe_from_words(z, w);
    for (int k = 0; k < reps; k++) { fe_invert_fermat(r, z); z = r; z.v[0] += 1; }
    fe_to_words(w, z);
    for (int i = 0; i < 8; i++) out[threadIdx.x * 8 + i] = w[i];
}
__global__ void __launch_bounds__(64) k_inv_safegcd(u32* out, const u32* in, int reps)
{
    u32 w[8];
    for (int i = 0; i < 8; i++) w[i] = in[threadIdx.x * 8 + i];
    fe z, r;
    fe_from_words(z, w);
    for (int k = 0; k < reps; k++) { fe_invert_safegcd(r, z); z = r; z.v[0] += 1; }
    fe_to_words(w, z);
    for (int i = 0; i < 8; i++) out[threadIdx.x * 8 + i] = w[i];
}
