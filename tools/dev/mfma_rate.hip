// Dev probe: issue rate of independent v_mfma_f32_16x16x32_bf16 from ONE wave per SIMD, 3 A x NJ B register tile (the bblock k-step).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
template <int NJ, int THREADS>
__global__ __launch_bounds__(THREADS) void probe(unsigned long long* out, int n, float* sink) {
    bf16x8 a[3], b[NJ];
    for (int m = 0; m < 3; ++m) for (int i = 0; i < 8; ++i) a[m][i] = (__bf16)(threadIdx.x * 0.001f + i + m);
    for (int j = 0; j < NJ; ++j) for (int i = 0; i < 8; ++i) b[j][i] = (__bf16)(i * 0.5f + j);
    f32x4 c[3][NJ] = {};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int m = 0; m < 3; ++m)
#pragma unroll
            for (int j = 0; j < NJ; ++j) c[m][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[m], b[j], c[m][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int m = 0; m < 3; ++m) for (int j = 0; j < NJ; ++j) for (int e = 0; e < 4; ++e) s += c[m][j][e];
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if (s == 12345.f) sink[0] = s;
}
template <int NJ, int THREADS> void run(unsigned long long* d, float* sink, const char* name) {
    const int n = 20000;
    probe<NJ, THREADS><<<256, THREADS>>>(d, n, sink); hipDeviceSynchronize();
    probe<NJ, THREADS><<<256, THREADS>>>(d, n, sink); hipDeviceSynchronize();
    unsigned long long h[256]; hipMemcpy(h, d, 256 * 8, hipMemcpyDeviceToHost);
    printf("%s: %.2f clk per MFMA (wave 0 of workgroup 0)\n", name, (double)h[0] / ((double)n * 3 * NJ));
}
int main() {
    unsigned long long* d; float* sink; hipMalloc(&d, 256 * 8); hipMalloc(&sink, 4);
    run<4, 256>(d, sink, "3x4 tile, 1 wave per SIMD");
    run<5, 256>(d, sink, "3x5 tile, 1 wave per SIMD");
    run<4, 512>(d, sink, "3x4 tile, 2 waves per SIMD");
    run<2, 256>(d, sink, "3x2 tile, 1 wave per SIMD");
    return 0;
}
