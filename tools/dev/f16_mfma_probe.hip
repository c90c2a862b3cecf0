// Dev probe for the fp16-split engine (NOTES/design_history_r1_r5.md §10): what v_mfma_f32_32x32x16_f16 / v_mfma_f32_16x16x32_f16 do on gfx950 with
//   (a) fp16 SUBNORMAL inputs (lo = fp16(x - fp16(x)) is subnormal for |x| < 2^-3): kept or flushed?
//   (b) products whose exact value needs 22 significand bits: summed exactly into the fp32 accumulator or truncated on the way
//       (the MX fp8 instruction truncates 2^-13 below the largest product of a group: tools/dev/fp8_acc_probe.hip)?
//   (c) a small product against a large accumulator value (alignment truncation inside the K sum?)
//   (d) issue rate against the bf16 instruction of the same shape.
//   hipcc --offload-arch=gfx950 -O2 tools/dev/f16_mfma_probe.hip -o /tmp/f16_mfma_probe && /tmp/f16_mfma_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

// D[i][j] = C + sum_k A[i][k] B[k][j]; every A element = a_val, every B element = b_val except k = 0 where B = b0
__global__ void probe32(float a_val, float b_val, float b0, float c_val, float* out) {
    const int lane = threadIdx.x;
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)a_val; b[i] = (_Float16)b_val; }
    if (lane < 32) b[0] = (_Float16)b0;                          // lanes 0..31 hold K = 0..7
    f32x16 c;
    for (int i = 0; i < 16; ++i) c[i] = c_val;
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    out[lane] = c[0];
}
__global__ void probe16(float a_val, float b_val, float b0, float c_val, float* out) {
    const int lane = threadIdx.x;
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)a_val; b[i] = (_Float16)b_val; }
    if (lane < 16) b[0] = (_Float16)b0;
    f32x4 c = {c_val, c_val, c_val, c_val};
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    out[lane] = c[0];
}

template <int T>   // 0: f16, 1: bf16
__global__ __launch_bounds__(256) void rate(float* out, int iters) {
    f32x16 acc[6];
    for (int t = 0; t < 6; ++t) for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    f16x8 a, b; bf16x8 ab, bb;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(float)(threadIdx.x % 3); b[i] = (_Float16)1.f; ab[i] = (__bf16)(float)(threadIdx.x % 3); bb[i] = (__bf16)1.f; }
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int t = 0; t < 6; ++t) {
            if constexpr (T == 0) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[t], 0, 0, 0);
            else acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, acc[t], 0, 0, 0);
        }
    float s = 0.f;
    for (int t = 0; t < 6; ++t) s += acc[t][0];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

static float run32(float a, float b, float b0, float c, float* d) {
    float h[64];
    probe32<<<1, 64>>>(a, b, b0, c, d); hipMemcpy(h, d, 256, hipMemcpyDeviceToHost);
    return h[0];
}
static float run16(float a, float b, float b0, float c, float* d) {
    float h[64];
    probe16<<<1, 64>>>(a, b, b0, c, d); hipMemcpy(h, d, 256, hipMemcpyDeviceToHost);
    return h[0];
}

int main() {
    float* d;
    hipMalloc(&d, 1 << 22);
    // (a) subnormal A (2^-20 is an fp16 subnormal: min normal 2^-14), normal B: 16 products of 2^-20 * 2^10 = 2^-10 -> 2^-6
    const float sub = ldexpf(1.f, -20);
    printf("(a) subnormal fp16 input 2^-20 x 2^10, K = 16 / 32: 32x32x16 -> %.9g  16x16x32 -> %.9g   (kept: %.9g / %.9g, flushed: 0)\n",
           run32(sub, 1024.f, 1024.f, 0.f, d), run16(sub, 1024.f, 1024.f, 0.f, d), 16 * ldexpf(1.f, -10), 32 * ldexpf(1.f, -10));
    const float tiny = ldexpf(1.f, -24);    // the smallest fp16 subnormal
    printf("    smallest subnormal 2^-24 x 2^14: 32x32x16 -> %.9g   (kept: %.9g)\n", run32(tiny, 16384.f, 16384.f, 0.f, d), 16 * ldexpf(1.f, -10));
    // (b) (1 + 2^-10)^2 = 1 + 2^-9 + 2^-20 needs 21 bits; 16 of them = 16 + 2^-5 + 2^-16 (exact in fp32)
    const float q = 1.f + ldexpf(1.f, -10);
    const double exact_b = 16.0 * (double)q * (double)q;
    printf("(b) 16 x (1 + 2^-10)^2: 32x32x16 -> %.10g   exact %.10g   (diff %.3g)\n", run32(q, q, q, 0.f, d), exact_b, (double)run32(q, q, q, 0.f, d) - exact_b);
    // (c) one large product + 15 small ones: 2^10 * 1 + 15 * (2^-12 ... ) -- is the small part truncated against the large one?
    //     A = 1 everywhere; B[k=0] = 1024, other B = 2^-13 -> exact sum = 1024 + 15 * 2^-13 = 1024.001831...
    const float small = ldexpf(1.f, -13);
    const double exact_c = 1024.0 + 15.0 * (double)small;
    printf("(c) 1024 + 15 x 2^-13 inside one K block: 32x32x16 -> %.10g   exact %.10g (fp32-rounded %.10g)\n", run32(1.f, small, 1024.f, 0.f, d), exact_c, (float)exact_c);
    printf("    accumulator 1024 + 16 products of 2^-13: -> %.10g   exact %.10g\n", run32(1.f, small, small, 1024.f, d), 1024.0 + 16.0 * small);
    const float small2 = ldexpf(1.f, -16);
    printf("    accumulator 1024 + 16 products of 2^-16 (sum 2^-12, representable next to 1024: ulp 2^-13): -> %.10g   exact %.10g\n", run32(1.f, small2, small2, 1024.f, d), 1024.0 + 16.0 * small2);
    // (d) rate
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int T = 0; T < 2; ++T) {
        const int iters = 20000, blocks = 256 * 2;
        float ms = 0.f;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (T == 0) rate<0><<<blocks, 256>>>(d, iters); else rate<1><<<blocks, 256>>>(d, iters);
            hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        }
        const double flop = (double)blocks * 4 * iters * 6 * 2.0 * 32 * 32 * 16;
        printf("(d) %s 32x32x16: %.1f TFLOP/s (2 workgroups of 4 waves per CU)\n", T == 0 ? "f16 " : "bf16", flop / (ms * 1e-3) / 1e12);
    }
    return 0;
}
