// Dev probe (round 6): issue rate of INDEPENDENT v_mfma_f32_16x16x32_f16 / v_mfma_f32_32x32x16_f16 from one or two waves per SIMD, as
// inline assembly (tools/dev/mfma_rate.hip's loop is rewritten by hipcc with accumulator copies and dependent chains: its numbers are
// not pipe rates), bare and with the fused block's ratio of ds_read_b128 per MFMA.  Prints s_memtime ticks per MFMA seen by wave 0.
//   hipcc --offload-arch=gfx950 -O2 tools/dev/mfma_rate2.hip -o /tmp/mfma_rate2 && /tmp/mfma_rate2
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) _Float16 h8;
typedef __attribute__((ext_vector_type(4))) float f4;
typedef __attribute__((ext_vector_type(16))) float f16v;

template <int READS>      // READS ds_read_b128 per group of 6 MFMAs (the fused block at J = 2: 5)
__global__ void probe16(unsigned long long* out, int n, float* sink) {
    __shared__ __attribute__((aligned(16))) char lds[32768];
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f); }
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) reinterpret_cast<float*>(lds)[i] = (float)i;
    __syncthreads();
    f4 c0 = {}, c1 = {}, c2 = {}, c3 = {}, c4 = {}, c5 = {}, c6 = {}, c7 = {}, c8 = {}, c9 = {}, c10 = {}, c11 = {};
    f4 r0 = {}, r1 = {}, r2 = {}, r3 = {}, r4 = {};
    const unsigned addr = (unsigned)((threadIdx.x & 63) * 16);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; ++i) {
#define M(c) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
#define R(r, off) if (READS > (off)) asm volatile("ds_read_b128 %0, %1 offset:" #off "*1024" : "=v"(r) : "v"(addr));
        R(r0, 0) M(c0) R(r1, 1) M(c1) R(r2, 2) M(c2) R(r3, 3) M(c3) R(r4, 4) M(c4) M(c5)
        R(r0, 0) M(c6) R(r1, 1) M(c7) R(r2, 2) M(c8) R(r3, 3) M(c9) R(r4, 4) M(c10) M(c11)
        asm volatile("s_waitcnt lgkmcnt(0)");
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    f4 s = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7 + c8 + c9 + c10 + c11 + r0 + r1 + r2 + r3 + r4;
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if (s[0] + s[1] + s[2] + s[3] == 12345.f) sink[0] = s[0];
}
__global__ void probe32(unsigned long long* out, int n, float* sink) {
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f); }
    f16v c0 = {}, c1 = {}, c2 = {}, c3 = {}, c4 = {}, c5 = {};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; ++i) {
#define M32(c) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
        M32(c0) M32(c1) M32(c2) M32(c3) M32(c4) M32(c5)
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    f16v s = c0 + c1 + c2 + c3 + c4 + c5;
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if (s[0] == 12345.f) sink[0] = s[0];
}
int main() {
    unsigned long long* d; float* sink; hipMalloc(&d, 256 * 8); hipMalloc(&sink, 4);
    const int n = 20000;
    unsigned long long h[256];
    auto rep = [&](const char* name, double per) { hipDeviceSynchronize(); hipMemcpy(h, d, 256 * 8, hipMemcpyDeviceToHost); printf("%-58s %.2f ticks per MFMA (wave 0 of workgroup 0)\n", name, (double)h[0] / per); };
    for (int threads : {256, 512, 1024}) {
        char nm[128];
        probe16<0><<<256, threads>>>(d, n, sink); probe16<0><<<256, threads>>>(d, n, sink);
        snprintf(nm, sizeof nm, "16x16x32, bare, %d wave(s) per SIMD", threads / 256); rep(nm, (double)n * 12);
        probe16<5><<<256, threads>>>(d, n, sink); probe16<5><<<256, threads>>>(d, n, sink);
        snprintf(nm, sizeof nm, "16x16x32 + 5 ds_read_b128 per 6, %d wave(s) per SIMD", threads / 256); rep(nm, (double)n * 12);
        if (threads <= 512) {
            probe32<<<256, threads>>>(d, n, sink); probe32<<<256, threads>>>(d, n, sink);
            snprintf(nm, sizeof nm, "32x32x16, bare, %d wave(s) per SIMD", threads / 256); rep(nm, (double)n * 6);
        }
    }
    return 0;
}
