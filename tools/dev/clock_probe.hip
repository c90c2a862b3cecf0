// Dev probe: what does s_memtime count?  One wave per SIMD runs N independent v_mfma_f32_32x32x16_bf16 (32 shader clocks each
// when the pipe is full) and N v_mfma_f32_16x16x32_bf16; ticks per MFMA from s_memtime, tick rate from the HIP-event wall time.
// hipcc --offload-arch=gfx950 -O2 tools/dev/clock_probe.hip -o tools/scratch/clock_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
template <int SHAPE>
__global__ __launch_bounds__(256) void probe(unsigned long long* out, int n, float* sink) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(i * 0.5f); }
    f32x16 c32[4] = {}; f32x4 c16[8] = {};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; ++i) {
        if (SHAPE == 32) {
#pragma unroll
            for (int k = 0; k < 4; ++k) c32[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c32[k], 0, 0, 0);
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) c16[k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c16[k], 0, 0, 0);
        }
    }
    float s = 0;
    for (int k = 0; k < 4; ++k) s += c32[k][0];
    for (int k = 0; k < 8; ++k) s += c16[k][0];
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if (s == 12345.f) sink[0] = s;
}
int main() {
    unsigned long long* d; float* sink; hipMalloc(&d, 256 * 8); hipMalloc(&sink, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int shape : {32, 16}) for (int grid : {1, 256}) {
        const int n = 200000;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (shape == 32) probe<32><<<grid, 256>>>(d, n, sink); else probe<16><<<grid, 256>>>(d, n, sink);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long h[256]; hipMemcpy(h, d, grid * 8, hipMemcpyDeviceToHost);
        const double mf = (double)n * (shape == 32 ? 4 : 8);
        printf("%dx%d grid %3d: %.2f ticks per MFMA, %.3f ms wall -> %.1f ns per MFMA, tick rate %.3f GHz\n", shape, shape, grid, h[0] / mf, ms, ms * 1e6 / mf, h[0] / (ms * 1e6));
    }
    return 0;
}
