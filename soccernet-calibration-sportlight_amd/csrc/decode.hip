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
#include <cfloat>

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
#define LAUNCH(V, Q) hipLaunchKernelGGL((kp_decode_kernel<V, Q>), grid, block, lds, s, d_logp, C, h, w, img_h, img_w, d_out)
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
