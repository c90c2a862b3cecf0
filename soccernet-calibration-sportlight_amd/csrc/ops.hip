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
#include "x3.hpp"
#include <algorithm>
#include "ops.hpp"
#include "softmax_px.hpp"

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
                                                           int C, int H, int W, unsigned* __restrict__ nonfinite) {
    // nonfinite (or null) = the second word of the network's range flag {overflow, nonfinite} (hrnet.cpp d_range; only the split-fp16
    // engine has one): a NaN / infinite frame value bumps nonfinite[0], a finite one beyond fp16's 65504 -- which the stem's in-kernel
    // split would clamp silently (conv.hpp) -- bumps the overflow word nonfinite[-1].  ToTensor's frames are in [0, 1]; forward() takes any fp32 tensor
    constexpr int GE = Vec<T>::GE;
    bool bad = false, big = false;
    const unsigned hw = (unsigned)(H * W);
    const size_t n = blockIdx.y;                                   // one image per grid row: no per-element division
    for (unsigned r = blockIdx.x * 256u + threadIdx.x; r < hw; r += gridDim.x * 256u) {
        typename Vec<T>::type v;
#pragma unroll
        for (int c = 0; c < GE; ++c) {
            const float f = c < C ? x[(n * C + c) * hw + r] : 0.0f;
            bad = bad || !(__builtin_fabsf(f) <= 3.4028234663852886e38f);
            big = big || (__builtin_fabsf(f) > 65504.0f && __builtin_fabsf(f) <= 3.4028234663852886e38f);
            v[c] = (T)f;
        }
        *reinterpret_cast<typename Vec<T>::type*>(y + (n * hw + r) * GE) = v;
    }
    if (nonfinite != nullptr && bad) atomicAdd(nonfinite, 1u);
#if SNCAL_X3_F16
    if (nonfinite != nullptr && big) atomicAdd(nonfinite - 1, 1u);
#endif
}

// uint8 HWC frames (what cv2.imread hands to ToTensor, make_submit.py:62-66) -> NHWC T: ToTensor's float32 x / 255,
// then the same conversion as nchw_to_nhwc; 3 bytes per pixel read instead of 12
template <typename T>
__global__ __launch_bounds__(256) void u8hwc_to_nhwc_kernel(const unsigned char* __restrict__ x, T* __restrict__ y, size_t total) {
    constexpr int GE = Vec<T>::GE;
    for (size_t p = blockIdx.x * 256ull + threadIdx.x; p < total; p += (size_t)gridDim.x * 256) {
        typename Vec<T>::type v;
#pragma unroll
        for (int c = 0; c < GE; ++c) v[c] = (T)(c < 3 ? (float)x[p * 3 + c] / 255.0f : 0.0f);
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
    if constexpr (GE == 8) {
        // bf16: the kernel is VALU-bound (unpack + 6 multiply-adds per tap-channel).  v_perm_b32 pairs the same
        // channel of two taps and v_dot2_f32_bf16 applies both weights with fp32 accumulation: 4 instructions per
        // channel instead of 10; the four weights are rounded to bf16 like the taps they multiply.
        typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
        bf16x2 wt, wb;
        wt[0] = (__bf16)(lx.w0 * ly.w0); wt[1] = (__bf16)(lx.w1 * ly.w0);
        wb[0] = (__bf16)(lx.w0 * ly.w1); wb[1] = (__bf16)(lx.w1 * ly.w1);
        const uint4 a0 = __builtin_bit_cast(uint4, v00), a1 = __builtin_bit_cast(uint4, v01);
        const uint4 b0 = __builtin_bit_cast(uint4, v10), b1 = __builtin_bit_cast(uint4, v11);
        const unsigned ta0[4] = {a0.x, a0.y, a0.z, a0.w}, ta1[4] = {a1.x, a1.y, a1.z, a1.w};
        const unsigned tb0[4] = {b0.x, b0.y, b0.z, b0.w}, tb1[4] = {b1.x, b1.y, b1.z, b1.w};
#pragma unroll
        for (int pr = 0; pr < 4; ++pr) {
            const bf16x2 tl = __builtin_bit_cast(bf16x2, __builtin_amdgcn_perm(ta1[pr], ta0[pr], 0x05040100u));
            const bf16x2 th = __builtin_bit_cast(bf16x2, __builtin_amdgcn_perm(ta1[pr], ta0[pr], 0x07060302u));
            const bf16x2 bl = __builtin_bit_cast(bf16x2, __builtin_amdgcn_perm(tb1[pr], tb0[pr], 0x05040100u));
            const bf16x2 bh = __builtin_bit_cast(bf16x2, __builtin_amdgcn_perm(tb1[pr], tb0[pr], 0x07060302u));
            acc[2 * pr] = __builtin_amdgcn_fdot2_f32_bf16(tl, wt, acc[2 * pr], false);
            acc[2 * pr] = __builtin_amdgcn_fdot2_f32_bf16(bl, wb, acc[2 * pr], false);
            acc[2 * pr + 1] = __builtin_amdgcn_fdot2_f32_bf16(th, wt, acc[2 * pr + 1], false);
            acc[2 * pr + 1] = __builtin_amdgcn_fdot2_f32_bf16(bh, wb, acc[2 * pr + 1], false);
        }
    } else {
#pragma unroll
        for (int e = 0; e < GE; ++e) {
            const float top = (float)v00[e] * lx.w0 + (float)v01[e] * lx.w1;
            const float bot = (float)v10[e] * lx.w0 + (float)v11[e] * lx.w1;
            acc[e] += top * ly.w0 + bot * ly.w1;
        }
    }
}

// x / d by multiply-high with the host-computed reciprocal floor(2^32 / d) + 1 (exact while x * d < 2^32): the emulated
// integer division costs ~25 VALU instructions in 32 bits and several times that in 64 bits -- with six of them per
// 16-byte element this kernel was instruction-bound at 3 TB/s
__device__ __forceinline__ unsigned udiv_magic(unsigned x, unsigned shift, unsigned magic) { return magic == 0 ? x >> shift : __umulhi(x >> shift, magic); }

template <typename T>
__global__ __launch_bounds__(256) void upsample_add_kernel(UpsampleAddParams p) {
    constexpr int GE = Vec<T>::GE;
    typedef typename Vec<T>::type V;
    const unsigned cg = (unsigned)(p.C / GE);
    const unsigned per_img = (unsigned)(p.H * p.W) * cg;           // 16-byte elements of one image (< 2^31)
    const int n = blockIdx.y;
    const size_t img_pix = (size_t)n * p.H * p.W;
    // XCD-aware walk: workgroup b runs on XCD b % 8 (observed dispatch rule, used for speed only), and each XCD has its own L2.  Every
    // output row re-reads the two source rows above / below it, so XCD x takes one contiguous BAND of rows of every image -- the
    // re-reads then hit that XCD's L2 instead of being fetched into all eight (the plain grid-stride walk interleaved the XCDs pixel by
    // pixel: with four sources the head's sum moved 17 x its output bytes through the fabric and ran at 1.4 TB/s)
    const unsigned xcd = blockIdx.x & 7u, j = blockIdx.x >> 3, nj = gridDim.x >> 3;          // gridDim.x is a multiple of 8
    const unsigned row_lo = (unsigned)p.H * xcd / 8u, row_hi = (unsigned)p.H * (xcd + 1u) / 8u;
    const unsigned e_hi = row_hi * (unsigned)p.W * cg;
    [[maybe_unused]] float amax = 0.f;                // fp32 path with a twin output: range tracker (x3.hpp)
    for (unsigned i = row_lo * (unsigned)p.W * cg + j * 256u + threadIdx.x; i < e_hi; i += nj * 256u) {
        const unsigned pl = udiv_magic(i, p.cg_shift, p.cg_magic);
        const int c0 = (int)(i - pl * cg) * GE;
        const unsigned uy = udiv_magic(pl, p.w_shift, p.w_magic);
        const int oy = (int)uy, ox = (int)(pl - uy * (unsigned)p.W);
        const size_t pix = img_pix + pl;
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
        if constexpr (GE == 4) {
            x3_track(amax, (float)o[0], (float)o[1]); x3_track(amax, (float)o[2], (float)o[3]);      // every output of the fp32 path (x3.hpp)
            if (p.out_twin) {          // bf16x3: hi = bf16(y), lo = bf16(y - hi) for the two-team convolution that reads this sum
                x3h4 th, tl;
#pragma unroll
                for (int e = 0; e < 4; ++e) X3_SPLIT1((float)o[e], th[e], tl[e]);
                char* const tw = reinterpret_cast<char*>(p.out_twin) + (pix * p.C + (c0 & ~15)) * 4 + (c0 & 15) * 2;
                *reinterpret_cast<x3h4*>(tw) = th;
                *reinterpret_cast<x3h4*>(tw + 32) = tl;
            }
            if (!p.out) continue;
        }
        *reinterpret_cast<V*>(reinterpret_cast<T*>(p.out) + pix * p.out_cstride + p.out_coff + c0) = o;
    }
    if constexpr (GE == 4) x3_report(amax, p.range);
}

// One workgroup = 64 consecutive pixels x up to 64 channels.  Reads are row-contiguous (4 lanes x 64 B per
// pixel), the channel reduction is thread-local + a 4-lane shuffle, and the NHWC->NCHW transpose goes
// through LDS so that every store instruction writes 256 contiguous bytes of one channel plane.
__global__ __launch_bounds__(256) void softmax_nchw_kernel(const float* __restrict__ logits, int cstride, int C,
                                                           size_t npix_total, size_t hw, int log_mode,
                                                           float* __restrict__ out) {
    __shared__ float s_t[64][65];
    const int t = threadIdx.x, px = t >> 2, q = t & 3;
    for (size_t p0 = (size_t)blockIdx.x * 64; p0 < npix_total; p0 += (size_t)gridDim.x * 64) {
        const size_t p = p0 + px;
        float v[16], rv[16];
        load_px16(logits, p, cstride, C, q, p < npix_total, v);
        softmax_px16(v, q, C, log_mode, rv);
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 16; ++j) s_t[q * 16 + j][px] = rv[j];
        __syncthreads();
        const int lp = t & 63;
        const size_t pp = p0 + lp;
        if (pp < npix_total) {
            const size_t n = pp / hw, r = pp - n * hw;
            for (int c = t >> 6; c < C; c += 4) out[(n * C + c) * hw + r] = s_t[c][lp];
        }
    }
}

static inline int grid_for(size_t items) {
    size_t b = (items + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}

int launch_nchw_to_nhwc(int dtype, const float* x, void* y, int N, int C, int H, int W, hipStream_t s, unsigned* nonfinite) {
    const size_t total = (size_t)N * H * W;
    if (dtype == SNCAL_BF16)
        SNCAL_LAUNCH(nchw_to_nhwc_kernel<__bf16>, dim3(grid_for((size_t)H * W), N), dim3(256), 0, s, x, (__bf16*)y, N, C, H, W, nonfinite);
    else
        SNCAL_LAUNCH(nchw_to_nhwc_kernel<float>, dim3(grid_for((size_t)H * W), N), dim3(256), 0, s, x, (float*)y, N, C, H, W, nonfinite);
    SNCAL_CHECK_LAUNCH();
    return SNCAL_OK;
}

int launch_u8hwc_to_nhwc(int dtype, const unsigned char* x, void* y, int N, int H, int W, hipStream_t s) {
    const size_t total = (size_t)N * H * W;
    if (dtype == SNCAL_BF16)
        SNCAL_LAUNCH(u8hwc_to_nhwc_kernel<__bf16>, dim3(grid_for(total)), dim3(256), 0, s, x, (__bf16*)y, total);
    else
        SNCAL_LAUNCH(u8hwc_to_nhwc_kernel<float>, dim3(grid_for(total)), dim3(256), 0, s, x, (float*)y, total);
    SNCAL_CHECK_LAUNCH();
    return SNCAL_OK;
}

int launch_upsample_add(int dtype, const UpsampleAddParams& p0, hipStream_t s) {
    UpsampleAddParams p = p0;
    const int ge = dtype == SNCAL_BF16 ? 8 : 4;
    const unsigned cg = (unsigned)(p.C / ge);
    const size_t per_img = (size_t)p.H * p.W * cg;
    // x / d = (x >> shift) / odd with d = odd << shift; multiply-high division by m = floor((2^32 - 1) / odd) + 1 is exact for every
    // y with y * (m * odd - 2^32) < 2^32 (the error term is < odd, 0 for odd = 1: magic 0 = shift only): checked for
    // (element index, C / GE) and (pixel index, W) -- 196 channel groups x 540 x 960 pixels fail without the shift, pass with it
    auto split = [](unsigned d, unsigned& shift, unsigned& magic, unsigned long long xmax) {
        shift = 0;
        while (d > 1 && (d & 1u) == 0) { d >>= 1; ++shift; }
        magic = d <= 1 ? 0u : 0xFFFFFFFFu / d + 1u;
        if (d <= 1) return true;
        const unsigned long long err = (unsigned long long)magic * d - (1ull << 32);
        return (xmax >> shift) * err < (1ull << 32);
    };
    const bool ok_c = split(cg, p.cg_shift, p.cg_magic, per_img), ok_w = split((unsigned)p.W, p.w_shift, p.w_magic, (unsigned long long)p.H * p.W);
    if (per_img >= (1ull << 31) || !ok_c || !ok_w) {
        set_error("upsample_add: image of %d x %d x %d is too large", p.H, p.W, p.C);
        return SNCAL_ERR_ARG;
    }
    const dim3 grid((unsigned)std::max<size_t>(8, std::min<size_t>((per_img / 8 + 255) / 256, 256) * 8), (unsigned)p.N);
    if (dtype == SNCAL_BF16)
        SNCAL_LAUNCH(upsample_add_kernel<__bf16>, grid, dim3(256), 0, s, p);
    else
        SNCAL_LAUNCH(upsample_add_kernel<float>, grid, dim3(256), 0, s, p);
    SNCAL_CHECK_LAUNCH();
    return SNCAL_OK;
}

int launch_softmax_nchw(const float* logits, int cstride, int C, size_t npix_total, size_t hw, int log_mode,
                        float* out, hipStream_t s) {
    if (C > 64 || cstride % 4 != 0) { set_error("softmax head supports at most 64 classes"); return SNCAL_ERR_ARG; }
    const size_t groups = (npix_total + 63) / 64;
    SNCAL_LAUNCH(softmax_nchw_kernel, dim3((unsigned)(groups > 16384 ? 16384 : groups)), dim3(256), 0, s, logits,
                       cstride, C, npix_total, hw, log_mode, out);
    SNCAL_CHECK_LAUNCH();
    return SNCAL_OK;
}

}  // namespace sncal
