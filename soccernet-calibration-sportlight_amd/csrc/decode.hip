// Heatmap decodes on gfx950: D1 keypoint decode and L2 line 2-peak decode.
//
// D1 follows HRNetPredictionTransform.__call__ (/root/reference/src/models/hrnet/transforms.py:228-239):
//   p = exp(logp); x = argmax_W(max_H p); y = argmax_H(max_W p); conf = min of the two maxima.
// One 256-thread workgroup owns one (b,c) plane and streams it from HBM exactly once: each of the 4
// waves walks whole rows (lane = float4 column group), so the per-row maximum is one wave reduction
// and the per-column maxima live in registers for the whole sweep.  Because exp is monotone, the
// reductions run on logp; exp_ref is applied only to the h row-maxima and w column-maxima, which is
// where the reference's "first occurrence after exp" tie rule is then evaluated exactly.
// HBM-bound: algorithmic bytes = B*C*h*w*4 read (+ B*(C-1)*12 written).
//
// L2 follows EHMPredictionTransform.mask_heat_points_gauss
// (/root/reference/src/models/line/transforms.py:224-280): relu, flat argmax, Gaussian suppression
// around the first peak, second flat argmax.  One workgroup per (b,c) plane, two sweeps (the second
// one is served by L2: a 135x240 plane is 130 KB).
#include "common.hpp"
#include "ops.hpp"
#include "softmax_px.hpp"
#include <cfloat>
#include <cstdlib>

namespace {

__device__ __forceinline__ float exp_ref(float x) {
    // float32(exp(float64(x))): the build's definition of the reference's torch.exp (oracle/decode.py)
    return (float)exp((double)x);
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ int wave_min_i(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o, 64));
    return v;
}

// block-wide (256 threads) reductions through a 4-entry LDS scratch
__device__ __forceinline__ float block_max(float v, float* scratch) {
    v = wave_max(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
    __syncthreads();
    return fmaxf(fmaxf(scratch[0], scratch[1]), fmaxf(scratch[2], scratch[3]));
}
__device__ __forceinline__ int block_min_i(int v, int* scratch) {
    v = wave_min_i(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
    __syncthreads();
    return min(min(scratch[0], scratch[1]), min(scratch[2], scratch[3]));
}

// first index i in [0,n) with exp_ref(v[i]) == max_j exp_ref(v[j]); also returns that maximum
__device__ __forceinline__ void first_max_after_exp(const float* v, int n, float* scratch, int* best_idx,
                                                    float* best_val) {
    float pm = -1.0f;   // exp() >= 0
    for (int i = threadIdx.x; i < n; i += 256) pm = fmaxf(pm, exp_ref(v[i]));
    pm = block_max(pm, scratch);
    int idx = 0x7fffffff;
    for (int i = threadIdx.x; i < n; i += 256)
        if (exp_ref(v[i]) == pm) { idx = i; break; }
    idx = block_min_i(idx, reinterpret_cast<int*>(scratch));
    *best_idx = idx;
    *best_val = pm;
}

template <int VEC, int NQ>
__global__ __launch_bounds__(256) void kp_decode_kernel(const float* __restrict__ logp, int C, int h, int w,
                                                        int img_h, int img_w, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int plane = blockIdx.x;
    const int b = plane / C, c = plane - b * C;
    if (c == C - 1) return;                      // background channel is dropped (transforms.py:238)
    float* s_row = smem;                          // [h]
    float* s_colw = smem + h;                     // [4][w]
    float* s_col = s_colw + 4 * w;                // [w]
    float* s_scr = s_col + w;                     // [4]
    const float* base = logp + (size_t)plane * h * w;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int nvec = (w + VEC - 1) / VEC;

    float cm[NQ][VEC];
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
        for (int e = 0; e < VEC; ++e) cm[q][e] = -INFINITY;

    for (int y = wave; y < h; y += 4) {
        const float* row = base + (size_t)y * w;
        float rm = -INFINITY;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int v = lane + 64 * q;
            if (v < nvec) {
                if constexpr (VEC == 4) {
                    const float4 t = *reinterpret_cast<const float4*>(row + 4 * v);
                    cm[q][0] = fmaxf(cm[q][0], t.x); cm[q][1] = fmaxf(cm[q][1], t.y);
                    cm[q][2] = fmaxf(cm[q][2], t.z); cm[q][3] = fmaxf(cm[q][3], t.w);
                    rm = fmaxf(rm, fmaxf(fmaxf(t.x, t.y), fmaxf(t.z, t.w)));
                } else {
                    const float t = row[v];
                    cm[q][0] = fmaxf(cm[q][0], t);
                    rm = fmaxf(rm, t);
                }
            }
        }
        rm = wave_max(rm);
        if (lane == 0) s_row[y] = rm;
    }
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int v = lane + 64 * q;
        if (v < nvec) {
#pragma unroll
            for (int e = 0; e < VEC; ++e)
                if (v * VEC + e < w) s_colw[wave * w + v * VEC + e] = cm[q][e];
        }
    }
    __syncthreads();
    for (int x = threadIdx.x; x < w; x += 256)
        s_col[x] = fmaxf(fmaxf(s_colw[x], s_colw[w + x]), fmaxf(s_colw[2 * w + x], s_colw[3 * w + x]));
    __syncthreads();

    int xi, yi;
    float xp, yp;
    first_max_after_exp(s_col, w, s_scr, &xi, &xp);   // x_prob, x = max_W(max_H p)   (:231)
    __syncthreads();
    first_max_after_exp(s_row, h, s_scr, &yi, &yp);   // y_prob, y = max_H(max_W p)   (:232)
    if (threadIdx.x == 0) {
        float* o = out + ((size_t)b * (C - 1) + c) * 3;
        o[0] = (float)((long long)xi * img_w) / (float)w;     // int64 * W / w -> true division, fp32
        o[1] = (float)((long long)yi * img_h) / (float)h;
        o[2] = fminf(xp, yp);
    }
}

// ---------------------------------------------------------------------------------------------
// Fused log-softmax + D1 decode for callers that do not want the heatmap (predict() / the pipeline): the
// (B,C,h,w) fp32 log-probabilities -- 30 MB per frame at 270x480 -- are never written or re-read.
//   pass 1  logsoftmax_rowcol_kernel  one workgroup per (strip of RS rows, frame): per 64-pixel tile the same
//           per-pixel channel softmax as softmax_nchw_kernel (softmax_px.hpp: bit-identical values), transposed
//           through LDS; a thread then owns (channel, 16 pixels): per-column maxima over the strip stay in registers,
//           per-row maxima accumulate in LDS.  Writes row maxima (B,C-1,h) and per-strip column maxima
//           (B,strips,C-1,w): 3 % of the heatmap's bytes.
//   pass 2  kp_finish_kernel          one workgroup per (b,c): column maxima over strips, then exactly the
//           first-occurrence-after-exp rule of kp_decode_kernel.
// max is exact in any grouping, so the keypoints equal softmax_nchw_kernel -> kp_decode_kernel bit for bit.
// ---------------------------------------------------------------------------------------------
constexpr int RC_ROWS = 18;      // rows per strip: 6 / 9 cost more in partials, 30 / 45 leave too few workgroups (0.65 ms at 18 vs 0.72-0.89)

__global__ __launch_bounds__(256) void logsoftmax_rowcol_kernel(const float* __restrict__ logits, int cstride, int C, int h, int w,
                                                                float* __restrict__ rowmax, float* __restrict__ colpart, int nstrips) {
    __shared__ float s_t[64][65];
    __shared__ float s_rm[4][64];
    __shared__ float s_row[RC_ROWS][64];
    const int t = threadIdx.x, px = t >> 2, q = t & 3;
    const int c = t & 63, qq = t >> 6;
    const int strip = blockIdx.x, b = blockIdx.y;
    const int y0 = strip * RC_ROWS, rows = min(RC_ROWS, h - y0);
    const int C1 = C - 1;
    for (int i = t; i < RC_ROWS * 64; i += 256) s_row[i >> 6][i & 63] = -INFINITY;
    const float* img = logits + (size_t)b * h * w * cstride;
    for (int x0 = 0; x0 < w; x0 += 64) {
        float cm[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) cm[i] = -INFINITY;
        const bool live = x0 + px < w;
        float v[16];
        sncal::load_px16(img, (size_t)y0 * w + x0 + px, cstride, C, q, live, v);
        for (int yl = 0; yl < rows; ++yl) {
            float r[16], vn[16];
            const bool more = yl + 1 < rows;                   // next row's pixel: in flight under this row's arithmetic
            sncal::load_px16(img, (size_t)(y0 + yl + (more ? 1 : 0)) * w + x0 + px, cstride, C, q, live && more, vn);
            sncal::softmax_px16(v, q, C, 1, r);
            __syncthreads();                                   // previous tile's readers are done with s_t / s_rm
#pragma unroll
            for (int j = 0; j < 16; ++j) s_t[q * 16 + j][px] = live ? r[j] : -INFINITY;
            __syncthreads();
            float rm = -INFINITY;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float u = s_t[c][qq * 16 + i];
                cm[i] = fmaxf(cm[i], u);
                rm = fmaxf(rm, u);
            }
            s_rm[qq][c] = rm;
            __syncthreads();
            if (t < 64) s_row[yl][t] = fmaxf(s_row[yl][t], fmaxf(fmaxf(s_rm[0][t], s_rm[1][t]), fmaxf(s_rm[2][t], s_rm[3][t])));
#pragma unroll
            for (int j = 0; j < 16; ++j) v[j] = vn[j];
        }
        if (c < C1) {
            float* dst = colpart + (((size_t)b * nstrips + strip) * C1 + c) * w + x0 + qq * 16;
#pragma unroll
            for (int i = 0; i < 16; ++i)
                if (x0 + qq * 16 + i < w) dst[i] = cm[i];
        }
    }
    __syncthreads();
    for (int i = t; i < rows * C1; i += 256) {
        const int cc = i / rows, yl = i - cc * rows;
        rowmax[((size_t)b * C1 + cc) * h + y0 + yl] = s_row[yl][cc];
    }
}

__global__ __launch_bounds__(256) void kp_finish_kernel(const float* __restrict__ rowmax, int nrparts, const float* __restrict__ colpart, int nstrips,
                                                        int C1, int h, int w, int img_h, int img_w, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* s_row = smem;                          // [h]
    float* s_col = smem + h;                      // [w]
    float* s_scr = s_col + w;                     // [4]
    const int plane = blockIdx.x, b = plane / C1, c = plane - b * C1;
    for (int y = threadIdx.x; y < h; y += 256) {          // row maxima: nrparts partial maxima per row (1: final values)
        float m = -INFINITY;
        for (int k = 0; k < nrparts; ++k) m = fmaxf(m, rowmax[((size_t)plane * h + y) * nrparts + k]);
        s_row[y] = m;
    }
    for (int x = threadIdx.x; x < w; x += 256) {
        float m = -INFINITY;
        for (int s = 0; s < nstrips; ++s) m = fmaxf(m, colpart[(((size_t)b * nstrips + s) * C1 + c) * w + x]);
        s_col[x] = m;
    }
    __syncthreads();
    int xi, yi;
    float xp, yp;
    first_max_after_exp(s_col, w, s_scr, &xi, &xp);
    __syncthreads();
    first_max_after_exp(s_row, h, s_scr, &yi, &yp);
    if (threadIdx.x == 0) {
        float* o = out + (size_t)plane * 3;
        o[0] = (float)((long long)xi * img_w) / (float)w;
        o[1] = (float)((long long)yi * img_h) / (float)h;
        o[2] = fminf(xp, yp);
    }
}

// ---------------------------------------------------------------------------------------------
// L2 line decode
// ---------------------------------------------------------------------------------------------
struct Peak { float v; int idx; };

__device__ __forceinline__ Peak better(Peak a, Peak b) {   // torch.max(view(-1)): first occurrence
    return (b.v > a.v || (b.v == a.v && b.idx < a.idx)) ? b : a;
}
__device__ __forceinline__ Peak block_peak(Peak p, Peak* scratch) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        Peak q;
        q.v = __shfl_xor(p.v, o, 64);
        q.idx = __shfl_xor(p.idx, o, 64);
        p = better(p, q);
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = p;
    __syncthreads();
    return better(better(scratch[0], scratch[1]), better(scratch[2], scratch[3]));
}

__global__ __launch_bounds__(256) void line_decode_kernel(const float* __restrict__ heat, int h, int w, float sigma,
                                                          float scale, float* __restrict__ out) {
    __shared__ Peak scratch[4];
    const int plane = blockIdx.x;
    const float* base = heat + (size_t)plane * h * w;
    const int n = h * w;
    Peak p1{-1.0f, 0x7fffffff};                       // relu output is >= 0
    for (int i = threadIdx.x; i < n; i += 256) {
        const float v = fmaxf(base[i], 0.0f);
        if (v > p1.v) { p1.v = v; p1.idx = i; }        // i increases: keeps the first occurrence
    }
    p1 = block_peak(p1, scratch);
    const float x1 = (float)(p1.idx % w), y1 = (float)(p1.idx / w);
    const float two_s2 = (float)(2.0 * (double)sigma * (double)sigma);   // python: 2.0 * sigma ** 2
    Peak p2{-1.0f, 0x7fffffff};
    for (int i = threadIdx.x; i < n; i += 256) {
        const float v = fmaxf(base[i], 0.0f);
        const float dx = (float)(i % w) - x1, dy = (float)(i / w) - y1;
        const float t = (dx * dx + dy * dy) / two_s2;
        // exp(-t) < 2^-25 for t > 17.33  =>  1 - mask rounds to exactly 1.0f: skip the exp there
        const float keep = (t > 17.5f) ? 1.0f : (1.0f - exp_ref(-t));
        const float m = v * keep;
        if (m > p2.v) { p2.v = m; p2.idx = i; }
    }
    p2 = block_peak(p2, scratch);
    if (threadIdx.x == 0) {
        float* o = out + (size_t)plane * 6;
        o[0] = x1 * scale; o[1] = y1 * scale; o[2] = p1.v;
        o[3] = (float)(p2.idx % w) * scale; o[4] = (float)(p2.idx / w) * scale; o[5] = p2.v;
    }
}

}  // namespace

namespace sncal {

size_t logsoftmax_decode_scratch(int B, int C, int h, int w) {
    const int nstrips = (h + RC_ROWS - 1) / RC_ROWS;
    return ((size_t)B * (C - 1) * h + (size_t)B * nstrips * (C - 1) * w) * sizeof(float);
}

int launch_logsoftmax_decode(const float* logits, int cstride, int C, int B, int h, int w, int img_h, int img_w, float* scratch,
                             float* kpts, hipStream_t s) {
    if (C > 64 || C < 2 || cstride % 4 != 0) { set_error("fused log-softmax decode supports 2..64 classes"); return SNCAL_ERR_ARG; }
    if (w > 8192 || h > 8192) { set_error("fused log-softmax decode: heatmap %dx%d too large", h, w); return SNCAL_ERR_ARG; }
    const int nstrips = (h + RC_ROWS - 1) / RC_ROWS;
    float* rowmax = scratch;
    float* colpart = scratch + (size_t)B * (C - 1) * h;
    SNCAL_LAUNCH_FIRST(logsoftmax_rowcol_kernel, dim3(nstrips, B), dim3(256), 0, s, logits, cstride, C, h, w, rowmax, colpart, nstrips);
    SNCAL_CHECK_LAUNCH();
    SNCAL_LAUNCH_LAST(kp_finish_kernel, dim3(B * (C - 1)), dim3(256), (size_t)(h + w + 4) * sizeof(float), s, rowmax, 1, colpart, nstrips,
                       C - 1, h, w, img_h, img_w, kpts);
    SNCAL_CHECK_LAUNCH();
    return SNCAL_OK;
}

// second half of the decode when the head kernel itself produced the maxima (head32.hip, decode-fused form): row maxima in `row_parts`
// partials per row [B][C-1][h][row_parts], column maxima in `col_parts` strips [B][col_parts][C-1][w]
int launch_kp_finish(const float* rowpart, int row_parts, const float* colpart, int col_parts, int C, int B, int h, int w, int img_h, int img_w,
                     float* kpts, hipStream_t s) {
    SNCAL_LAUNCH(kp_finish_kernel, dim3(B * (C - 1)), dim3(256), (size_t)(h + w + 4) * sizeof(float), s, rowpart, row_parts, colpart, col_parts,
                 C - 1, h, w, img_h, img_w, kpts);
    SNCAL_CHECK_LAUNCH();
    return SNCAL_OK;
}

}  // namespace sncal

extern "C" int sncal_heatmap_decode(const float* d_logp, int B, int C, int h, int w, int img_h, int img_w,
                                    float* d_out, void* stream) {
    SNCAL_CHECK_ARG(B >= 0 && C >= 2 && h > 0 && w > 0, "sncal_heatmap_decode: bad shape B=%d C=%d h=%d w=%d", B, C, h, w);
    SNCAL_CHECK_ARG(w <= 2048 && h <= 8192, "sncal_heatmap_decode: heatmap %dx%d exceeds 8192x2048", h, w);
    if (B == 0) return SNCAL_OK;
    SNCAL_CHECK_ARG(d_logp && d_out, "sncal_heatmap_decode: null pointer");
    const size_t lds = (size_t)(h + 5 * w + 4) * sizeof(float);
    const bool vec4 = (w % 4 == 0) && ((reinterpret_cast<uintptr_t>(d_logp) & 15) == 0);
    const int nvec = vec4 ? w / 4 : w;
    const int nq = (nvec + 63) / 64;
    dim3 grid(B * C), block(256);
    hipStream_t s = sncal::as_stream(stream);
#define LAUNCH(V, Q) SNCAL_LAUNCH((kp_decode_kernel<V, Q>), grid, block, lds, s, d_logp, C, h, w, img_h, img_w, d_out)
    if (vec4) {
        if (nq <= 1) LAUNCH(4, 1); else if (nq <= 2) LAUNCH(4, 2); else if (nq <= 4) LAUNCH(4, 4); else LAUNCH(4, 8);
    } else {
        SNCAL_CHECK_ARG(nq <= 32, "sncal_heatmap_decode: width %d not a multiple of 4 is limited to 2048", w);
        if (nq <= 1) LAUNCH(1, 1); else if (nq <= 2) LAUNCH(1, 2); else if (nq <= 4) LAUNCH(1, 4);
        else if (nq <= 8) LAUNCH(1, 8); else if (nq <= 16) LAUNCH(1, 16); else LAUNCH(1, 32);
    }
#undef LAUNCH
    SNCAL_CHECK_LAUNCH();
    return SNCAL_OK;
}

extern "C" int sncal_line_decode(const float* d_heat, int B, int C, int h, int w, float sigma, float scale,
                                 float* d_out, void* stream) {
    SNCAL_CHECK_ARG(B >= 0 && C > 0 && h > 0 && w > 0 && sigma > 0, "sncal_line_decode: bad arguments");
    if (B == 0) return SNCAL_OK;
    SNCAL_CHECK_ARG(d_heat && d_out, "sncal_line_decode: null pointer");
    hipLaunchKernelGGL(line_decode_kernel, dim3(B * C), dim3(256), 0, sncal::as_stream(stream), d_heat, h, w, sigma,
                       scale, d_out);
    SNCAL_CHECK_LAUNCH();
    return SNCAL_OK;
}
