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

// The same per-pixel log-softmax for the accumulator layout of head32.hip: TWO lanes (l and l ^ 32, hi = l >> 5) hold one pixel, 32
// channels each -- v[8 * k + e] = channel 16 * k + 8 * hi + e, k = 0..3 (k = quarter q of softmax_px16).  Bit-identical to
// softmax_px16(log_mode = 1): the maximum is exact in any grouping; quarter k's sum runs 0 + e_0 + ... + e_15 in channel order, i.e.
// through the hi = 0 lane's eight terms and on through the hi = 1 lane's; the four quarter sums combine as (s0 + s1) + (s2 + s3).
__device__ __forceinline__ void logsoftmax_px32x2(const float (&v)[32], int hi, int C, float (&r)[32]) {
#pragma clang fp contract(off)      // scoped to this body: the summation order below is the contract, whatever the including file's FP mode
    float m = v[0];
#pragma unroll
    for (int j = 1; j < 32; ++j) m = fmaxf(m, v[j]);
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    float e[32], part[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#pragma unroll
        for (int j = 0; j < 8; ++j) e[8 * k + j] = (16 * k + 8 * hi + j < C) ? expf(v[8 * k + j] - m) : 0.f;
        const float lower = __shfl_xor(0.f + e[8 * k] + e[8 * k + 1] + e[8 * k + 2] + e[8 * k + 3] + e[8 * k + 4] + e[8 * k + 5] + e[8 * k + 6] + e[8 * k + 7], 32, 64);
        // hi = 1: continue the partner's partial sum with this lane's eight terms (hi = 0 computes a value nobody uses)
        part[k] = lower + e[8 * k] + e[8 * k + 1] + e[8 * k + 2] + e[8 * k + 3] + e[8 * k + 4] + e[8 * k + 5] + e[8 * k + 6] + e[8 * k + 7];
    }
    float ssum = (part[0] + part[1]) + (part[2] + part[3]);
    const float other = __shfl_xor(ssum, 32, 64);
    ssum = hi ? ssum : other;                       // the hi = 1 lane holds the pixel's sum
    const float ls = logf(ssum);
#pragma unroll
    for (int j = 0; j < 32; ++j) r[j] = (v[j] - m) - ls;
}

}  // namespace sncal
