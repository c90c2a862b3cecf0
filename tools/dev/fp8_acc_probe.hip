// How exactly does v_mfma_scale_f32_32x32x64_f8f6f4 (e4m3 x e4m3, unit block scales) accumulate?  (run on the GPU box)
//   hipcc --offload-arch=gfx950 -O2 tools/dev/fp8_acc_probe.hip -o /tmp/fp8_acc_probe && /tmp/fp8_acc_probe
// Every product of two e4m3 values is exact in fp32, so an fp32-exact accumulation would return the correctly rounded sum.
// The probe puts ONE large product (2^s) next to 63 small ones (t each) and next to a non-zero C input and prints the sum the
// instruction returns against the exact one: the smallest t / 2^s that still registers is the width of the internal adder.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

__global__ void k(const unsigned char* A, const unsigned char* B, float c0, float* D) {
    const int l = threadIdx.x;
    v8i a, b;
    for (int i = 0; i < 8; ++i) {
        a[i] = *(const int*)(A + ((l & 31) * 64 + (l >> 5) * 32 + i * 4));
        b[i] = *(const int*)(B + ((l & 31) * 64 + (l >> 5) * 32 + i * 4));
    }
    v16f acc;
    for (int r = 0; r < 16; ++r) acc[r] = c0;
    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = acc[r];
}

static unsigned char enc(float f) {            // exact powers of two / small integers only
    if (f == 0) return 0;
    int e; float m = std::frexp(std::fabs(f), &e);      // f = m 2^e, m in [0.5,1)
    int E = e - 1 + 7; int q = (int)std::lround((m * 2 - 1) * 8);
    if (E < 1) { q = (int)std::lround(std::fabs(f) * 512); E = 0; }
    return (unsigned char)((f < 0 ? 0x80 : 0) | (E << 3) | q);
}

int main() {
    unsigned char *dA, *dB; float* dD;
    hipMalloc(&dA, 2048); hipMalloc(&dB, 2048); hipMalloc(&dD, 4096);
    std::vector<float> D(1024);
    printf("one product 2^8 * 1 plus 63 products t * 1 (exact sum = 256 + 63 t), C = 0:\n");
    for (int sh = 0; sh <= 17; ++sh) {
        std::vector<unsigned char> A(2048), B(2048, enc(1.0f));
        // t = 2^-sh * 256 ... expressed as a * b with a = 2^p, b = 2^q inside e4m3's range
        const float t = std::ldexp(256.0f, -sh);
        int p = 0, q = 0; float tt = t;             // split the exponent between the operands
        while (tt < std::ldexp(1.0f, -6)) { tt *= 2; --q; }
        for (int m = 0; m < 32; ++m) for (int kk = 0; kk < 64; ++kk) { A[m * 64 + kk] = enc(kk == 0 ? 256.0f : tt); }
        for (int n = 0; n < 32; ++n) for (int kk = 0; kk < 64; ++kk) { B[n * 64 + kk] = enc(kk == 0 ? 1.0f : std::ldexp(1.0f, q)); }
        (void)p;
        hipMemcpy(dA, A.data(), 2048, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 2048, hipMemcpyHostToDevice);
        k<<<1, 64>>>(dA, dB, 0.0f, dD);
        hipMemcpy(D.data(), dD, 4096, hipMemcpyDeviceToHost);
        const double exact = 256.0 + 63.0 * t;
        printf("  t = 2^%-3d  got %.10g  exact %.10g  (fp32 of exact %.10g)  lost %.3g\n", 8 - sh, D[0], exact, (float)exact, exact - D[0]);
    }
    printf("same with the large term in C (C = 256, 64 products t):\n");
    for (int sh = 10; sh <= 26; sh += 2) {
        const float t = std::ldexp(256.0f, -sh);
        float tt = t; int q = 0;
        while (tt < std::ldexp(1.0f, -6)) { tt *= 2; --q; }
        std::vector<unsigned char> A(2048, enc(tt)), B(2048, enc(std::ldexp(1.0f, q)));
        hipMemcpy(dA, A.data(), 2048, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 2048, hipMemcpyHostToDevice);
        k<<<1, 64>>>(dA, dB, 256.0f, dD);
        hipMemcpy(D.data(), dD, 4096, hipMemcpyDeviceToHost);
        const double exact = 256.0 + 64.0 * t;
        printf("  t = 2^%-3d  got %.10g  exact %.10g  lost %.3g\n", 8 - sh, D[0], exact, exact - D[0]);
    }
    return 0;
}
