// Probe of v_mfma_scale_f32_32x32x64_f8f6f4 with e4m3 operands and unit block scales (run on the GPU box):
//   hipcc --offload-arch=gfx950 -O2 tools/dev/mx_probe.hip -o /tmp/mx_probe && /tmp/mx_probe
// Checks the operand layout assumed by conv_tt.hip: lane l holds 32 consecutive K bytes of row / column l & 31, K block l >> 5.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

__global__ void k(const unsigned char* A, const unsigned char* B, float* D, int scale) {
    const int l = threadIdx.x;
    v8i a, b;
    for (int i = 0; i < 8; ++i) {
        a[i] = *(const int*)(A + ((l & 31) * 64 + (l >> 5) * 32 + i * 4));     // A[m][k] row-major 32 x 64
        b[i] = *(const int*)(B + ((l & 31) * 64 + (l >> 5) * 32 + i * 4));     // B^T[n][k]
    }
    v16f acc = {0};
    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc, 0, 0, 0, scale, 0, scale);
    for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = acc[r];   // D[m][n]
}

static float fp8_to_f(unsigned char v) {
    const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
    float f = e == 0 ? std::ldexp((float)m, -9) : std::ldexp(1.0f + m / 8.0f, e - 7);
    return s ? -f : f;
}

int main() {
    std::vector<unsigned char> A(32 * 64), B(32 * 64);
    srand(1);
    for (auto& v : A) { v = rand() & 0xff; if ((v & 0x7f) == 0x7f) v = 0x38; }
    for (auto& v : B) { v = rand() & 0xff; if ((v & 0x7f) == 0x7f) v = 0x38; }
    unsigned char *dA, *dB; float* dD;
    hipMalloc(&dA, A.size()); hipMalloc(&dB, B.size()); hipMalloc(&dD, 32 * 32 * 4);
    hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size(), hipMemcpyHostToDevice);
    for (int scale : {0x7f7f7f7f, 0x7f, 0}) {
        k<<<1, 64>>>(dA, dB, dD, scale);
        std::vector<float> D(32 * 32);
        hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
        double maxrel = 0, ratio = 0;
        for (int m = 0; m < 32; ++m)
            for (int n = 0; n < 32; ++n) {
                double ref = 0;
                for (int kk = 0; kk < 64; ++kk) ref += (double)fp8_to_f(A[m * 64 + kk]) * fp8_to_f(B[n * 64 + kk]);
                maxrel = std::fmax(maxrel, std::fabs(D[m * 32 + n] - ref) / (std::fabs(ref) + 1e-3));
                if (m == 3 && n == 5) ratio = D[m * 32 + n] / ref;
            }
        printf("scale operand 0x%08x: max rel err %.3e   D[3][5]/ref = %g\n", scale, maxrel, ratio);
    }
    return 0;
}
