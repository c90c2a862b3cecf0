// Memory-bound helper kernels of the HRNet engine (NHWC, 16-byte channel groups per thread).
//   nchw_to_nhwc      input frames (B,3,H,W) fp32 -> NHWC T, channels zero-padded to one k-group
//   upsample_add      HighResolutionModule fuse: out = [relu](base + sum_s bilinear_up(src_s))
//                     (/root/reference/src/models/hrnet/hrnet.py:229-244, align_corners=True)
//   upsample_concat   head: bilinear_up(branch) written into a channel slice of the concat tensor
//                     (hrnet.py:489-509)
//   softmax_nchw      LogSoftmax / Softmax over channels of the NHWC fp32 logits -> NCHW fp32 heatmaps
//                     (hrnet.py:329, line/hrnet.py:101)
// All are HBM-bound streaming kernels: one 16-byte vector per lane, grid-stride.
#include "common.hpp"
#include "ops.hpp"

namespace sncal {

template <typename T> struct Vec;
template <> struct Vec<__bf16> {
    static constexpr int GE = 8;
    typedef __attribute__((ext_vector_type(8))) __bf16 type;
};
template <> struct Vec<float> {
    static constexpr int GE = 4;
    typedef __attribute__((ext_vector_type(4))) float type;
};

template <typename T>
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ x, T* __restrict__ y, int N,
                                                           int C, int H, int W) {
    constexpr int GE = Vec<T>::GE;
    const size_t total = (size_t)N * H * W;
    for (size_t p = blockIdx.x * 256ull + threadIdx.x; p < total; p += (size_t)gridDim.x * 256) {
        const size_t hw = (size_t)H * W;
        const size_t n = p / hw, r = p - n * hw;
        typename Vec<T>::type v;
#pragma unroll
        for (int c = 0; c < GE; ++c) v[c] = (T)(c < C ? x[(n * C + c) * hw + r] : 0.0f);
        *reinterpret_cast<typename Vec<T>::type*>(y + p * GE) = v;
    }
}

// PyTorch's align_corners=True source index: scale = (in-1)/(out-1) in fp32, src = scale*dst
struct Lerp { int i0, i1; float w0, w1; };
__device__ __forceinline__ Lerp lerp_idx(int o, int in_size, float scale) {
    const float src = scale * (float)o;
    Lerp l;
    l.i0 = (int)src;
    if (l.i0 > in_size - 1) l.i0 = in_size - 1;
    l.i1 = l.i0 + (l.i0 < in_size - 1 ? 1 : 0);
    l.w1 = src - (float)l.i0;
    l.w0 = 1.0f - l.w1;
    return l;
}

template <typename T>
__device__ __forceinline__ void bilinear_acc(float (&acc)[Vec<T>::GE], const T* src, int n, int Hs, int Ws, int C,
                                             int c0, const Lerp& ly, const Lerp& lx) {
    constexpr int GE = Vec<T>::GE;
    typedef typename Vec<T>::type V;
    const size_t base = (size_t)n * Hs * Ws;
    const V v00 = *reinterpret_cast<const V*>(src + (base + (size_t)ly.i0 * Ws + lx.i0) * C + c0);
    const V v01 = *reinterpret_cast<const V*>(src + (base + (size_t)ly.i0 * Ws + lx.i1) * C + c0);
    const V v10 = *reinterpret_cast<const V*>(src + (base + (size_t)ly.i1 * Ws + lx.i0) * C + c0);
    const V v11 = *reinterpret_cast<const V*>(src + (base + (size_t)ly.i1 * Ws + lx.i1) * C + c0);
#pragma unroll
    for (int e = 0; e < GE; ++e) {
        const float top = (float)v00[e] * lx.w0 + (float)v01[e] * lx.w1;
        const float bot = (float)v10[e] * lx.w0 + (float)v11[e] * lx.w1;
        acc[e] += top * ly.w0 + bot * ly.w1;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void upsample_add_kernel(UpsampleAddParams p) {
    constexpr int GE = Vec<T>::GE;
    typedef typename Vec<T>::type V;
    const int cg = p.C / GE;
    const size_t total = (size_t)p.N * p.H * p.W * cg;
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c0 = (int)(i % cg) * GE;
        size_t pix = i / cg;
        const int ox = (int)(pix % p.W);
        size_t t = pix / p.W;
        const int oy = (int)(t % p.H);
        const int n = (int)(t / p.H);
        float acc[GE];
        if (p.base) {
            const V b = *reinterpret_cast<const V*>(reinterpret_cast<const T*>(p.base) + pix * p.C + c0);
#pragma unroll
            for (int e = 0; e < GE; ++e) acc[e] = (float)b[e];
        } else {
#pragma unroll
            for (int e = 0; e < GE; ++e) acc[e] = 0.f;
        }
        for (int s = 0; s < p.nsrc; ++s) {
            const Lerp ly = lerp_idx(oy, p.Hs[s], p.sy[s]);
            const Lerp lx = lerp_idx(ox, p.Ws[s], p.sx[s]);
            bilinear_acc<T>(acc, reinterpret_cast<const T*>(p.src[s]), n, p.Hs[s], p.Ws[s], p.C, c0, ly, lx);
        }
        V o;
#pragma unroll
        for (int e = 0; e < GE; ++e) o[e] = (T)(p.relu ? fmaxf(acc[e], 0.f) : acc[e]);
        *reinterpret_cast<V*>(reinterpret_cast<T*>(p.out) + pix * p.out_cstride + p.out_coff + c0) = o;
    }
}

__global__ __launch_bounds__(256) void softmax_nchw_kernel(const float* __restrict__ logits, int cstride, int C,
                                                           size_t npix_total, size_t hw, int log_mode,
                                                           float* __restrict__ out) {
    for (size_t p = blockIdx.x * 256ull + threadIdx.x; p < npix_total; p += (size_t)gridDim.x * 256) {
        const float* row = logits + p * cstride;
        float m = -INFINITY;
        for (int c = 0; c < C; ++c) m = fmaxf(m, row[c]);
        float s = 0.f;
        for (int c = 0; c < C; ++c) s += expf(row[c] - m);
        const size_t n = p / hw, r = p - n * hw;
        float* o = out + n * C * hw + r;
        if (log_mode) {
            const float ls = logf(s);
            for (int c = 0; c < C; ++c) o[(size_t)c * hw] = (row[c] - m) - ls;
        } else {
            const float inv = 1.0f / s;
            for (int c = 0; c < C; ++c) o[(size_t)c * hw] = expf(row[c] - m) * inv;
        }
    }
}

static inline int grid_for(size_t items) {
    size_t b = (items + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}

int launch_nchw_to_nhwc(int dtype, const float* x, void* y, int N, int C, int H, int W, hipStream_t s) {
    const size_t total = (size_t)N * H * W;
    if (dtype == SNCAL_BF16)
        hipLaunchKernelGGL(nchw_to_nhwc_kernel<__bf16>, dim3(grid_for(total)), dim3(256), 0, s, x, (__bf16*)y, N, C, H, W);
    else
        hipLaunchKernelGGL(nchw_to_nhwc_kernel<float>, dim3(grid_for(total)), dim3(256), 0, s, x, (float*)y, N, C, H, W);
    SNCAL_CHECK_LAUNCH();
    return SNCAL_OK;
}

int launch_upsample_add(int dtype, const UpsampleAddParams& p, hipStream_t s) {
    const int ge = dtype == SNCAL_BF16 ? 8 : 4;
    const size_t total = (size_t)p.N * p.H * p.W * (p.C / ge);
    if (dtype == SNCAL_BF16)
        hipLaunchKernelGGL(upsample_add_kernel<__bf16>, dim3(grid_for(total)), dim3(256), 0, s, p);
    else
        hipLaunchKernelGGL(upsample_add_kernel<float>, dim3(grid_for(total)), dim3(256), 0, s, p);
    SNCAL_CHECK_LAUNCH();
    return SNCAL_OK;
}

int launch_softmax_nchw(const float* logits, int cstride, int C, size_t npix_total, size_t hw, int log_mode,
                        float* out, hipStream_t s) {
    hipLaunchKernelGGL(softmax_nchw_kernel, dim3(grid_for(npix_total)), dim3(256), 0, s, logits, cstride, C,
                       npix_total, hw, log_mode, out);
    SNCAL_CHECK_LAUNCH();
    return SNCAL_OK;
}

}  // namespace sncal
