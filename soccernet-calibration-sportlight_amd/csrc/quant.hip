// fp8 (OCP e4m3) activation helpers of the C5 path (BASELINE config 5: "HRNet-W48 fp8 (CDNA4 fp8 MFMA) 1920x1080"):
//   absmax_bf16_kernel    calibration: per-tensor max |x| of a bf16 activation tensor (one atomicMax per workgroup)
//   quantize_fp8_kernel   bf16 [N][H][W][C] -> fp8 e4m3 twin, x / scale, saturated to +-448 (no NaN encodings are produced)
// Both are plain HBM streams (2 B read / 1 B written per element, 16-byte vector accesses).  The twins feed the fp8
// variant of the two-team convolution kernel (conv_tt.hip); convolutions that follow another fp8 convolution get their
// twin written by the producer's epilogue and never pass through here.
// The reference has no reduced-precision path (predict() is fp32: src/models/hrnet/metamodel.py:127-134); this is the
// build's own C5 arithmetic, judged by the tolerance sweep against the fp32 engine (tests/test_fp8_gpu.py).
#include "common.hpp"
#include "x3.hpp"
#include "ops.hpp"

namespace sncal {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

__global__ __launch_bounds__(256) void absmax_bf16_kernel(const bf16x8* __restrict__ x, size_t n8, unsigned* __restrict__ out) {
    float m = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) {
        const bf16x8 v = x[i];
#pragma unroll
        for (int e = 0; e < 8; ++e) m = fmaxf(m, fabsf((float)v[e]));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    __shared__ float part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmaxf(fmaxf(part[0], part[1]), fmaxf(part[2], part[3]));
        atomicMax(out, __float_as_uint(m));                 // non-negative floats order like their bit patterns
    }
}

__device__ __forceinline__ unsigned pack4_fp8(float a, float b, float c, float d) {
    int r = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
    r = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, r, true);
    return (unsigned)r;
}

__global__ __launch_bounds__(256) void quantize_fp8_kernel(const bf16x8* __restrict__ x, uint2* __restrict__ y, size_t n8, float inv_scale) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) {
        const bf16x8 v = x[i];
        float f[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = fminf(fmaxf((float)v[e] * inv_scale, -448.f), 448.f);
        y[i] = make_uint2(pack4_fp8(f[0], f[1], f[2], f[3]), pack4_fp8(f[4], f[5], f[6], f[7]));
    }
}

// fp32 [N][H][W][C] -> split twin for the split-arithmetic convolutions (conv_tt.hip MODE 2): per pixel and 16-channel group
// [16 hi | 16 lo] 16-bit codes with hi = rne16(x), lo = rne16(x - hi) (x3.hpp).  One thread = 8 channels: 32 B read, 16 B + 16 B written.
__global__ __launch_bounds__(256) void split_f32_kernel(const float4* __restrict__ x, x3h8* __restrict__ y, size_t n8, unsigned* __restrict__ range) {
    float amax = 0.f;                                  // range tracker (x3.hpp)
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) {
        const float4 a = x[2 * i], b = x[2 * i + 1];
        const float f[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        x3h8 h, l;
#pragma unroll
        for (int e = 0; e < 8; ++e) X3_SPLIT1(f[e], h[e], l[e]);
#pragma unroll
        for (int e = 0; e < 8; e += 2) x3_track(amax, f[e], f[e + 1]);
        const size_t g16 = i >> 1, half = i & 1;            // 16-channel group, which 8 of its channels
        y[g16 * 4 + half] = h;
        y[g16 * 4 + 2 + half] = l;
    }
    x3_report(amax, range);
}

int launch_split_f32(const void* x, void* y, size_t n, hipStream_t s, unsigned* range) {
    if (n % 16) { set_error("split: element count %zu is not a multiple of 16", n); return SNCAL_ERR_ARG; }
    const size_t n8 = n / 8;
    const unsigned blocks = (unsigned)std::min<size_t>((n8 + 255) / 256, 4096);
    SNCAL_LAUNCH(split_f32_kernel, dim3(blocks), dim3(256), 0, s, reinterpret_cast<const float4*>(x), reinterpret_cast<x3h8*>(y), n8, range);
    SNCAL_CHECK_LAUNCH();
    return SNCAL_OK;
}

int launch_absmax_bf16(const void* x, size_t n, unsigned* d_out, hipStream_t s) {
    if (n % 8) { set_error("absmax: element count %zu is not a multiple of 8", n); return SNCAL_ERR_ARG; }
    const size_t n8 = n / 8;
    const unsigned blocks = (unsigned)std::min<size_t>((n8 + 255) / 256, 2048);
    SNCAL_LAUNCH(absmax_bf16_kernel, dim3(blocks), dim3(256), 0, s, reinterpret_cast<const bf16x8*>(x), n8, d_out);
    SNCAL_CHECK_LAUNCH();
    return SNCAL_OK;
}

int launch_quantize_fp8(const void* x, void* y, size_t n, float scale, hipStream_t s) {
    if (n % 8) { set_error("quantize: element count %zu is not a multiple of 8", n); return SNCAL_ERR_ARG; }
    const size_t n8 = n / 8;
    const unsigned blocks = (unsigned)std::min<size_t>((n8 + 255) / 256, 2048);
    SNCAL_LAUNCH(quantize_fp8_kernel, dim3(blocks), dim3(256), 0, s, reinterpret_cast<const bf16x8*>(x), reinterpret_cast<uint2*>(y), n8, 1.0f / scale);
    SNCAL_CHECK_LAUNCH();
    return SNCAL_OK;
}

}  // namespace sncal
