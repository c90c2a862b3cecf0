// CU / queue interference probe (tools/dev/holder_probe.py): k single-wave workgroups that spin for `ticks` of the 100 MHz clock.
// variant 0: a handful of registers; 1: the whole register file of a SIMD (256 VGPRs + 256 AGPRs, like a camera-solve wavefront);
// 2: like 1, four waves per workgroup (a whole CU, like calibrate_kernel)
#include <hip/hip_runtime.h>
__global__ void hold_light(unsigned long long ticks, unsigned* sink) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    unsigned n = 0;
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) { __builtin_amdgcn_s_sleep(8); ++n; }
    if (n == 0xffffffffu) *sink = n;
}
template <int BLOCK>
__global__ __launch_bounds__(BLOCK, 1) void hold_fat(unsigned long long ticks, unsigned* sink) {
    asm volatile("v_mov_b32 v250, 0\n v_accvgpr_write_b32 a250, 0" ::: "v250", "a250");
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    unsigned n = 0;
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) { __builtin_amdgcn_s_sleep(8); ++n; }
    if (n == 0xffffffffu) *sink = n;
}
// busy variant: fp64 FMA chain (no sleep), like a Levenberg-Marquardt crawl
__global__ __launch_bounds__(64, 1) void hold_busy(unsigned long long ticks, unsigned* sink) {
    asm volatile("v_mov_b32 v250, 0\n v_accvgpr_write_b32 a250, 0" ::: "v250", "a250");
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    double a = 1.0 + threadIdx.x, b = 1.0000001;
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) {
#pragma unroll
        for (int i = 0; i < 64; ++i) a = a * b + 1e-9;
    }
    if (a == 0.123) *sink = 1;
}
// variant 4: like hold_busy plus a private array the compiler must keep in scratch memory (dynamic index): 4.6 KB per lane, touched rarely
__global__ __launch_bounds__(64, 1) void hold_scratch(unsigned long long ticks, unsigned* sink, int idx) {
    asm volatile("v_mov_b32 v250, 0\n v_accvgpr_write_b32 a250, 0" ::: "v250", "a250");
    double priv[576];
    for (int i = 0; i < 576; ++i) priv[i] = i * 0.5 + threadIdx.x;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    double a = 1.0 + threadIdx.x, b = 1.0000001;
    int j = idx;
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) {
#pragma unroll
        for (int i = 0; i < 64; ++i) a = a * b + 1e-9;
        a += priv[j % 576]; j = (j * 7 + 1) & 1023;          // one scratch load per ~300 clk
    }
    if (a == 0.123) *sink = 1;
}
extern "C" int holder_launch(int variant, int k, double ms, void* stream) {
    static unsigned* sink = nullptr;
    if (!sink) hipMalloc(&sink, 4);
    const unsigned long long ticks = (unsigned long long)(ms * 1e5);
    hipStream_t s = (hipStream_t)stream;
    if (variant == 0) hipLaunchKernelGGL(hold_light, dim3(k), dim3(64), 0, s, ticks, sink);
    else if (variant == 1) hipLaunchKernelGGL(hold_fat<64>, dim3(k), dim3(64), 0, s, ticks, sink);
    else if (variant == 2) hipLaunchKernelGGL(hold_fat<256>, dim3(k), dim3(256), 0, s, ticks, sink);
    else if (variant == 3) hipLaunchKernelGGL(hold_busy, dim3(k), dim3(64), 0, s, ticks, sink);
    else if (variant == 4) hipLaunchKernelGGL(hold_scratch, dim3(k), dim3(64), 0, s, ticks, sink, 3);
    else {              // 5: like the voter's task launch -- 960 single-wave workgroups of which only the first k stay
        hipLaunchKernelGGL(hold_scratch, dim3(k), dim3(64), 0, s, ticks, sink, 3);
    }
    return (int)hipGetLastError();
}
