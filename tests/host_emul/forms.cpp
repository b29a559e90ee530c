// tests/host_emul/forms.cpp -- TEST INFRASTRUCTURE: the alternative field representations of
// tools/ubench/field_forms.cuh compiled for the host (same C model of the primitives as emul.cpp), so that the A/B the
// GPU microbenchmark times is between CORRECT implementations: each form's field ops against big integers (via
// canonical words) and a whole X25519 through each form's ladder step.
#include "field_forms.cuh"

using namespace c25519;
namespace c25519 { unsigned long long emul_mad_overflows = 0; }

extern "C" {

// op: 0 mul, 1 sqr, 2 add, 3 sub, 4 a + 121665 b, 5 invert.  form: 8 or 9.  32-byte little-endian operands mod p.
void forms_fe_op(unsigned char* out, const unsigned char* a, const unsigned char* b, size_t n, int op, int form)
{
    for (size_t i = 0; i < n; i++) {
        u32 aw[8], bw[8], ow[8];
        memcpy(aw, a + 32 * i, 32);
        memcpy(bw, b + 32 * i, 32);
        if (form == 8) {
            fe8 x, y, r;
            fe8_from_words(x, aw); fe8_from_words(y, bw);
            switch (op) {
            case 0: fe8_mul(r, x, y); break;
            case 1: fe8_sqr(r, x); break;
            case 2: fe8_add(r, x, y); break;
            case 3: fe8_sub(r, x, y); break;
            case 4: fe8_mul121665_add(r, x, y); break;
            default: fe8_invert(r, x); break;
            }
            fe8_to_words(ow, r);
        } else {
            fe9 x, y, r, t;
            fe9_from_words(x, aw); fe9_from_words(y, bw);
            switch (op) {
            case 0: fe9_mul(r, x, y); break;
            case 1: fe9_sqr(r, x); break;
            case 2: fe9_add(r, x, y); break;
            case 3: fe9_sub(t, x, y); fe9_carry(r, t); break;
            case 4: fe9_mul121665_add(r, x, y); break;
            default: fe9_invert(r, x); break;
            }
            fe9_to_words(ow, r);
        }
        memcpy(out + 32 * i, ow, 32);
    }
}

void forms_x25519(unsigned char* out, const unsigned char* pk, const unsigned char* sk, size_t n, int form)
{
    for (size_t i = 0; i < n; i++) {
        u32 u[8], k[8], w[8];
        memcpy(u, pk + 32 * i, 32);
        memcpy(k, sk + 32 * i, 32);
        if (form == 8) x25519_form<fe8>(w, u, k);
        else x25519_form<fe9>(w, u, k);
        memcpy(out + 32 * i, w, 32);
    }
}

unsigned long long forms_overflows(void) { return emul_mad_overflows; }

}  // extern "C"
