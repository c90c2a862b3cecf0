// Per-pixel channel softmax shared by softmax_nchw_kernel (ops.hip) and the fused log-softmax + keypoint decode
// (decode.hip): ONE definition, so that both paths produce bit-identical values.
// Four consecutive lanes (q = lane & 3) hold one pixel, 16 channels each (channel = q * 16 + j; invalid ones = -inf).
#pragma once
#include <hip/hip_runtime.h>

namespace sncal {

__device__ __forceinline__ void softmax_px16(const float (&v)[16], int q, int C, int log_mode, float (&r)[16]) {
    float m = v[0];
#pragma unroll
    for (int j = 1; j < 16; ++j) m = fmaxf(m, v[j]);
    m = fmaxf(m, __shfl_xor(m, 1, 64));
    m = fmaxf(m, __shfl_xor(m, 2, 64));
    float e[16], ssum = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) { e[j] = (q * 16 + j < C) ? expf(v[j] - m) : 0.f; ssum += e[j]; }
    ssum += __shfl_xor(ssum, 1, 64);
    ssum += __shfl_xor(ssum, 2, 64);
    const float ls = logf(ssum), inv = 1.0f / ssum;
#pragma unroll
    for (int j = 0; j < 16; ++j) r[j] = log_mode ? (v[j] - m) - ls : e[j] * inv;
}

// loads the 16 channels of lane-quarter q of pixel p (NHWC fp32, channel stride cstride, multiple of 4)
__device__ __forceinline__ void load_px16(const float* __restrict__ logits, size_t p, int cstride, int C, int q, bool live,
                                          float (&v)[16]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float4 f = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
        const int c = q * 16 + j * 4;
        if (live && c < cstride) f = *reinterpret_cast<const float4*>(logits + p * cstride + c);
        v[j * 4 + 0] = c + 0 < C ? f.x : -INFINITY; v[j * 4 + 1] = c + 1 < C ? f.y : -INFINITY;
        v[j * 4 + 2] = c + 2 < C ? f.z : -INFINITY; v[j * 4 + 3] = c + 3 < C ? f.w : -INFINITY;
    }
}

}  // namespace sncal
