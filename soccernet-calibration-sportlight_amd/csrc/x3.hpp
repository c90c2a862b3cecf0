// The 16-bit type of the split (x3) arithmetic: every fp32 operand x = hi + lo, hi = rne16(x), lo = rne16(x - hi), every product
// w.x = w_hi.x_hi + w_hi.x_lo + w_lo.x_hi on the 16-bit matrix pipe with fp32 accumulation (NOTES/design_history_r1_r5.md §9.3 / 10).
//   SNCAL_X3_F16 = 1 (default since round 4): hi and lo are IEEE fp16 -- 11 + 11 significand bits, the dropped lo.lo term is ~2^-22 of a
//     product; v_mfma_*_f16 keeps fp16 subnormals and accumulates exactly (tools/dev/f16_mfma_probe.hip), so small operands lose
//     precision gracefully (|x| < 2^-3: lo is subnormal, absolute error <= 2^-25) and nothing needs scaling; |x| is clamped to the
//     fp16 range (65504: a network whose activations leave it has left any sane operating point);
//   SNCAL_X3_F16 = 0: bf16 -- 8 + 8 bits, dropped term ~2^-16, fp32's exponent range (rounds 3's engine, `bf16x3`).
// Same instruction rate, same bytes: the two differ in the conversions only.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstring>

#ifndef SNCAL_X3_F16
#define SNCAL_X3_F16 1
#endif

namespace sncal {

#if SNCAL_X3_F16
typedef _Float16 x3h;
#define SNCAL_X3_NAME "fp16x3"
#define X3_MFMA_32x32x16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)
#define X3_MFMA_16x16x32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0)
#else
typedef __bf16 x3h;
#define SNCAL_X3_NAME "bf16x3"
#define X3_MFMA_32x32x16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)
#define X3_MFMA_16x16x32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)
#endif
typedef __attribute__((ext_vector_type(8))) x3h x3h8;
typedef __attribute__((ext_vector_type(4))) x3h x3h4;

// split: hi = rne16(v), lo = rne16(v - hi).  fp16: |v| is clamped to the format's range FIRST -- a stale or padded operand far outside it
// (it meets a zero weight, e.g. a gather lane outside its box) must stay finite in BOTH parts: with only hi clamped, lo = v - 65504
// converted to +inf and inf x 0 poisoned the sum (found by the batch-independence test: the poison depends on what the workspace held).
// Two spellings of the clamp: fminf(fmaxf()) compiles to a canonicalising v_max plus a v_med3, the intrinsic to the v_med3 alone.  Which
// one is faster is a matter of the surrounding schedule (measured round 4: the intrinsic gains 4 % in the head and 1.6 % in the generic
// stride-2 kernel and LOSES 1 % in the fused block and 0.3 % in the two-team kernel), so the call site chooses: X3_SPLIT / X3_SPLIT1.
__device__ __forceinline__ float x3_clamp(float v) {
#if SNCAL_X3_F16
    return __builtin_fminf(__builtin_fmaxf(v, -65504.f), 65504.f);
#else
    return v;
#endif
}
__device__ __forceinline__ float x3_clamp1(float v) {
#if SNCAL_X3_F16
    return __builtin_amdgcn_fmed3f(v, -65504.f, 65504.f);
#else
    return v;
#endif
}
// ReLU as ONE instruction (fmaxf(v, 0) is a canonicalising v_max plus the v_max itself when v comes out of an MFMA), and ReLU + clamp
// as one: the operand of X3_SPLIT_RAW
__device__ __forceinline__ float x3_relu(float v) { return __builtin_amdgcn_fmed3f(v, 0.f, __builtin_inff()); }
__device__ __forceinline__ float x3_relu_clamp(float v) {
#if SNCAL_X3_F16
    return __builtin_amdgcn_fmed3f(v, 0.f, 65504.f);
#else
    return x3_relu(v);
#endif
}
// (macros: vector elements do not bind to references)
#define X3_SPLIT1(v, hi, lo) do { const float x3v_ = ::sncal::x3_clamp1(v); (hi) = (::sncal::x3h)x3v_; (lo) = (::sncal::x3h)(x3v_ - (float)(hi)); } while (0)
#define X3_SPLIT_RAW(v, hi, lo) do { const float x3v_ = (v); (hi) = (::sncal::x3h)x3v_; (lo) = (::sncal::x3h)(x3v_ - (float)(hi)); } while (0)
#define X3_SPLIT(v, hi, lo) do { const float x3v_ = ::sncal::x3_clamp(v); (hi) = (::sncal::x3h)x3v_; (lo) = (::sncal::x3h)(x3v_ - (float)(hi)); } while (0)

// Two / eight values at once.  fp16 splits: per pair two v_med3_f32 (the clamp; with RELU the lower bound is 0, i.e. ReLU rides along),
// one v_cvt_pk_f16_f32 (the hi pair) and the lo pair as v_fma_mixlo_f16 / v_fma_mixhi_f16 -- D.f16 = f16(-hi.f16 * 1.0 + x.f32) reads the
// packed hi halves directly, so the unpack (v_cvt_f32_f16 x 2), the v_pk_add_f32 and the second v_cvt_pk of the compiler's code for
// lo = (f16)(x - (float)hi) are two instructions: 5 VALU instructions per pair instead of 7 (9 with a separate ReLU).  x - hi is exact in
// fp32 (both are multiples of ulp32(x), |x - hi| <= ulp16(x) / 2), so the fused form rounds the same number once: identical bits on 2^24
// values incl. fp16 subnormals, the clamp boundary, infinities and signed zeros (tools/dev/fma_mix_probe.hip).  Epilogues are VALU work of a
// LOADING wave beside a multiplying partner (~6-14 clk per instruction there): conv_tt_body.inc, bblockx3.hip, headx3.hip, bneckx3.hip.
// `lob` = x3_lower(relu): the lower clamp bound as a run-time (wave-uniform) operand, so that one copy of an epilogue serves layers with
// and without ReLU.
__device__ __forceinline__ float x3_lower(bool relu) {
#if SNCAL_X3_F16
    return relu ? 0.f : -65504.f;
#else
    return relu ? 0.f : -__builtin_inff();
#endif
}
__device__ __forceinline__ void x3_split2(float a, float b, float lob, unsigned& hi, unsigned& lo) {
#if SNCAL_X3_F16
    typedef float f2_t __attribute__((ext_vector_type(2)));
    typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
    const float x0 = __builtin_amdgcn_fmed3f(a, lob, 65504.f), x1 = __builtin_amdgcn_fmed3f(b, lob, 65504.f);
    const f2_t xx = {x0, x1};
    hi = __builtin_bit_cast(unsigned, __builtin_convertvector(xx, h2_t));
    unsigned d = 0;
    asm("v_fma_mixlo_f16 %0, -%1, 1.0, %2 op_sel_hi:[1,0,0]" : "+v"(d) : "v"(hi), "v"(x0));
    asm("v_fma_mixhi_f16 %0, -%1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(d) : "v"(hi), "v"(x1));
    lo = d;
#else
    typedef x3h h2_t __attribute__((ext_vector_type(2)));
    h2_t h, l;
    const float x0 = __builtin_amdgcn_fmed3f(a, lob, __builtin_inff()), x1 = __builtin_amdgcn_fmed3f(b, lob, __builtin_inff());
    X3_SPLIT_RAW(x0, h[0], l[0]);
    X3_SPLIT_RAW(x1, h[1], l[1]);
    hi = __builtin_bit_cast(unsigned, h); lo = __builtin_bit_cast(unsigned, l);
#endif
}
// The way back: hi + lo of element `half` (0 / 1) of two packed words as ONE instruction -- v_fma_mix_f32 D = lo.f16 * 1.0 + hi.f16, the
// fp32 sum of two exactly converted halves, i.e. (float)lo + (float)hi bit for bit (3 instructions as the compiler writes it).
template <int HALF>
__device__ __forceinline__ float x3_join(unsigned hi_pk, unsigned lo_pk) {
#if SNCAL_X3_F16
    float d;
    if constexpr (HALF == 0) asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel_hi:[1,0,1]" : "=v"(d) : "v"(lo_pk), "v"(hi_pk));
    else asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[1,0,1] op_sel_hi:[1,0,1]" : "=v"(d) : "v"(lo_pk), "v"(hi_pk));
    return d;
#else
    return __uint_as_float(HALF ? lo_pk & 0xffff0000u : lo_pk << 16) + __uint_as_float(HALF ? hi_pk & 0xffff0000u : hi_pk << 16);
#endif
}
// Run-time range flag (round 5).  The clamp above is silent by construction; a network whose activations really leave +-65504 -- the
// reference's fp32 predict() has no such limit (src/models/hrnet/metamodel.py:127-134) -- must not come back with plausible heatmaps.
// Every kernel that splits activations keeps the largest |value| it has split in one register (v_max3_f32 with |.| source modifiers:
// half an instruction per value) and, when that exceeds the fp16 range, bumps a sticky per-network counter as it leaves
// (sncal_hrnet_range_status).  |v| also counts a large NEGATIVE pre-activation that the ReLU would have zeroed anyway: a false alarm
// in the safe direction.  Infinities count; a NaN cannot arise from finite weights (checked at finalize) and finite frames (checked by the
// layout kernel) without an overflow first.  The bf16 build of the engine has fp32's exponent range: nothing to track.
__device__ __forceinline__ void x3_track(float& amax, float a, float b) {
#if SNCAL_X3_F16 && !defined(SNCAL_X3_NO_TRACK)      // (-DSNCAL_X3_NO_TRACK: A/B builds that price the tracker)
    amax = __builtin_fmaxf(__builtin_fmaxf(amax, __builtin_fabsf(a)), __builtin_fabsf(b));
#endif
}
__device__ __forceinline__ void x3_track1(float& amax, float a) {
#if SNCAL_X3_F16 && !defined(SNCAL_X3_NO_TRACK)
    amax = __builtin_fmaxf(amax, __builtin_fabsf(a));
#endif
}
__device__ __forceinline__ void x3_report(float amax, unsigned* range) {
#if SNCAL_X3_F16
    if (range != nullptr && amax > 65504.f) atomicAdd(range, 1u);
#endif
}
typedef unsigned x3u4 __attribute__((ext_vector_type(4)));
typedef unsigned x3u2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void x3_split8(const float (&v)[8], float lob, x3u4& hi, x3u4& lo) {
#pragma unroll
    for (int k = 0; k < 4; ++k) { unsigned h, l; x3_split2(v[2 * k], v[2 * k + 1], lob, h, l); hi[k] = h; lo[k] = l; }
}
__device__ __forceinline__ void x3_split4(const float (&v)[4], float lob, x3u2& hi, x3u2& lo) {
#pragma unroll
    for (int k = 0; k < 2; ++k) { unsigned h, l; x3_split2(v[2 * k], v[2 * k + 1], lob, h, l); hi[k] = h; lo[k] = l; }
}

// ... and the same with the range tracker (above) fed
__device__ __forceinline__ void x3_split8(const float (&v)[8], float lob, x3u4& hi, x3u4& lo, float& amax) {
#pragma unroll
    for (int k = 0; k < 4; ++k) x3_track(amax, v[2 * k], v[2 * k + 1]);
    x3_split8(v, lob, hi, lo);
}
__device__ __forceinline__ void x3_split4(const float (&v)[4], float lob, x3u2& hi, x3u2& lo, float& amax) {
    x3_track(amax, v[0], v[1]); x3_track(amax, v[2], v[3]);
    x3_split4(v, lob, hi, lo);
}

// host side (weight packing): the same two codes
inline uint16_t x3_code_host(float v) {
#if SNCAL_X3_F16
    v = v > 65504.f ? 65504.f : v < -65504.f ? -65504.f : v;
    const _Float16 h = (_Float16)v;
    uint16_t c; memcpy(&c, &h, 2); return c;
#else
    uint32_t u; memcpy(&u, &v, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
#endif
}
inline float x3_value_host(uint16_t c) {
#if SNCAL_X3_F16
    _Float16 h; memcpy(&h, &c, 2); return (float)h;
#else
    const uint32_t u = (uint32_t)c << 16; float f; memcpy(&f, &u, 4); return f;
#endif
}
inline void x3_split_host(float v, uint16_t* hi, uint16_t* lo) {
#if SNCAL_X3_F16
    v = v > 65504.f ? 65504.f : v < -65504.f ? -65504.f : v;
#endif
    *hi = x3_code_host(v); *lo = x3_code_host(v - x3_value_host(*hi));
}

}  // namespace sncal
