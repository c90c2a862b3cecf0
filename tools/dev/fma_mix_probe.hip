// Probe (gfx950): the fp16 split with v_fma_mixlo/hi_f16 (x3.hpp x3_split2) against the plain C split on 2^24 values: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -w
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split2_new(float a, float b, unsigned& hi, unsigned& lo, bool relu) {
    const float x0 = relu ? __builtin_amdgcn_fmed3f(a, 0.f, 65504.f) : __builtin_amdgcn_fmed3f(a, -65504.f, 65504.f);
    const float x1 = relu ? __builtin_amdgcn_fmed3f(b, 0.f, 65504.f) : __builtin_amdgcn_fmed3f(b, -65504.f, 65504.f);
    const f2 xx = {x0, x1};
    const h2 hh = __builtin_convertvector(xx, h2);
    hi = __builtin_bit_cast(unsigned, hh);
    unsigned d = 0;
    asm("v_fma_mixlo_f16 %0, -%1, 1.0, %2 op_sel_hi:[1,0,0]" : "+v"(d) : "v"(hi), "v"(x0));
    asm("v_fma_mixhi_f16 %0, -%1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(d) : "v"(hi), "v"(x1));
    lo = d;
}
__device__ __forceinline__ void split1_old(float v, _Float16& hi, _Float16& lo, bool relu) {
    if (relu) v = fmaxf(v, 0.f);
    const float x = __builtin_fminf(__builtin_fmaxf(v, -65504.f), 65504.f);
    hi = (_Float16)x; lo = (_Float16)(x - (float)hi);
}
__global__ void k(const float* in, unsigned* out_new, unsigned* out_old, int n, int relu) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * i + 1 >= n) return;
    unsigned hi, lo;
    split2_new(in[2 * i], in[2 * i + 1], hi, lo, relu);
    out_new[2 * i] = hi; out_new[2 * i + 1] = lo;
    h2 oh, ol; _Float16 a, b;
    split1_old(in[2 * i], a, b, relu); oh[0] = a; ol[0] = b;
    split1_old(in[2 * i + 1], a, b, relu); oh[1] = a; ol[1] = b;
    out_old[2 * i] = __builtin_bit_cast(unsigned, oh); out_old[2 * i + 1] = __builtin_bit_cast(unsigned, ol);
}
int main() {
    const int n = 1 << 24;
    std::vector<float> h(n);
    srand(1);
    for (int i = 0; i < n; ++i) {
        unsigned u = ((unsigned)rand() << 16) ^ (unsigned)rand() ^ ((unsigned)rand() << 30);
        if (i % 4 == 0) { float f; memcpy(&f, &u, 4); h[i] = std::isnan(f) ? 1.f : f; }           // any bit pattern (no NaN)
        else if (i % 4 == 1) h[i] = ldexpf((float)rand() / RAND_MAX - 0.5f, rand() % 40 - 30);   // the activations' range
        else if (i % 4 == 2) h[i] = ldexpf((float)rand() / RAND_MAX - 0.5f, rand() % 10 - 30);   // fp16-subnormal range
        else h[i] = (float)(rand() % 200001 - 100000);
    }
    const float sp[] = {0.f, -0.f, 65504.f, -65504.f, 65519.99f, 65520.f, 1e30f, -1e30f, INFINITY, -INFINITY, 5.96e-8f, 2.98e-8f, 1e-45f, 6.1e-5f, 1.0009765625f, 1.00048828125f};
    for (unsigned i = 0; i < sizeof(sp) / 4; ++i) h[i] = sp[i];
    float* d; unsigned *a, *b;
    hipMalloc(&d, n * 4); hipMalloc(&a, n * 4); hipMalloc(&b, n * 4);
    hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
    for (int relu = 0; relu < 2; ++relu) {
        k<<<n / 2 / 256, 256>>>(d, a, b, n, relu);
        std::vector<unsigned> ha(n), hb(n);
        hipMemcpy(ha.data(), a, n * 4, hipMemcpyDeviceToHost); hipMemcpy(hb.data(), b, n * 4, hipMemcpyDeviceToHost);
        long bad = 0, zsign = 0;
        for (int i = 0; i < n; ++i) if (ha[i] != hb[i]) {
            // differences in the sign of a zero half only?
            unsigned x = ha[i] ^ hb[i];
            bool only_zero_sign = true;
            for (int s = 0; s < 32; s += 16) { unsigned xa = (ha[i] >> s) & 0xffff, xb = (hb[i] >> s) & 0xffff; if (xa != xb && !((xa & 0x7fff) == 0 && (xb & 0x7fff) == 0)) only_zero_sign = false; }
            (void)x;
            if (only_zero_sign) ++zsign; else { if (bad < 8) printf("relu %d i %d in %g %g: new %08x old %08x\n", relu, i, h[i & ~1], h[i | 1], ha[i], hb[i]); ++bad; }
        }
        printf("relu %d: %d words, mismatches %ld (zero-sign-only differences %ld)\n", relu, n, bad, zsign);
    }
    return 0;
}
