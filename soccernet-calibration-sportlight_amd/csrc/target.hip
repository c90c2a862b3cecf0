// Training-target synthesis (heatmap half): HRNetLoss.create_target = create_heatmaps + background channel
//   /root/reference/src/models/hrnet/loss.py:7-52 (gaussian, create_heatmaps), :81-87 (create_target)
// keypoints (B,N,3) fp32 [x, y, visibility] in heatmap pixels -> (B,N+1,h,w) fp32:
//   channel n = exp(-((x - mu_x)/sigma)^2 / 2) * exp(-((y - mu_y)/sigma)^2 / 2) where the point is "visible", else 0
//   channel N = 1 - max over the N keypoint channels
// "visible" is the reference's own test, loss.py:49: any(keypoints == 1, dim=-1) over ALL three components (a point whose
// x or y is exactly 1.0 counts as visible even with flag 0; mirrored, not fixed).
// Pure HBM-write work: (N+1)*h*w*4 bytes per frame (30 MB at 58 x 270 x 480).  A thread owns one column x of a 32-row
// strip: its N column Gaussians live in registers, the strip's N x 32 row Gaussians in LDS, so the exp count is
// N*(w + h) per frame instead of N*w*h, and every store instruction writes 1 KB of one channel row.
// The arithmetic is fp32 step by step as torch evaluates it; exp is float32(exp(float64)), i.e. correctly rounded
// (torch's CPU exp is within 1 ulp of that: tests/test_target_gpu.py).
#include "common.hpp"
#include "../../include/sncal.h"

namespace {

constexpr int TG_ROWS = 32, TG_MAXN = 64;

__device__ __forceinline__ float gauss1(float x, float mu, float sigma) {
    const float d = (x - mu) / sigma;                 // torch.div(x - mu, sigma)
    return (float)exp((double)(-(d * d) / 2.0f));     // exp(-(d ** 2) / 2.0)
}

__global__ __launch_bounds__(256) void create_target_kernel(const float* __restrict__ kp, int N, float sigma, int h, int w,
                                                            float* __restrict__ out) {
    __shared__ float s_gy[TG_MAXN][TG_ROWS];
    __shared__ float s_kp[TG_MAXN][3];
    __shared__ int s_vis[TG_MAXN];
    const int t = threadIdx.x, b = blockIdx.z, y0 = blockIdx.y * TG_ROWS, x = blockIdx.x * 256 + t;
    const int rows = min(TG_ROWS, h - y0);
    for (int i = t; i < N * 3; i += 256) s_kp[i / 3][i % 3] = kp[((size_t)b * N) * 3 + i];
    __syncthreads();
    for (int i = t; i < N; i += 256) s_vis[i] = (s_kp[i][0] == 1.0f) | (s_kp[i][1] == 1.0f) | (s_kp[i][2] == 1.0f);
    for (int i = t; i < N * TG_ROWS; i += 256) {
        const int n = i / TG_ROWS, r = i - n * TG_ROWS;
        s_gy[n][r] = gauss1((float)(y0 + r), s_kp[n][1], sigma);
    }
    __syncthreads();
    if (x >= w) return;
    float gx[TG_MAXN];
#pragma unroll
    for (int n = 0; n < TG_MAXN; ++n) gx[n] = (n < N && s_vis[n]) ? gauss1((float)x, s_kp[n][0], sigma) : 0.f;
    const size_t plane = (size_t)h * w;
    float* const o = out + (size_t)b * (N + 1) * plane + (size_t)y0 * w + x;
    for (int r = 0; r < rows; ++r) {
        float m = -INFINITY;
#pragma unroll
        for (int n = 0; n < TG_MAXN; ++n) {
            if (n < N) {
                const float v = s_vis[n] ? gx[n] * s_gy[n][r] : 0.f;
                o[(size_t)n * plane + (size_t)r * w] = v;
                m = fmaxf(m, v);
            }
        }
        o[(size_t)N * plane + (size_t)r * w] = 1.0f - m;
    }
}

}  // namespace

extern "C" int sncal_create_target(const float* d_kpts, int B, int N, float sigma, int h, int w, float* d_out, void* stream) {
    SNCAL_CHECK_ARG(B >= 0 && N > 0 && N <= TG_MAXN && h > 0 && w > 0, "sncal_create_target: B=%d N=%d h=%d w=%d (N <= %d)", B, N, h, w, TG_MAXN);
    SNCAL_CHECK_ARG(sigma > 0.f, "sncal_create_target: sigma %g", (double)sigma);
    if (B == 0) return SNCAL_OK;
    SNCAL_CHECK_ARG(d_kpts && d_out, "sncal_create_target: null pointer");
    SNCAL_CHECK_ARG(B <= 65535 && (h + TG_ROWS - 1) / TG_ROWS <= 65535, "sncal_create_target: grid too large");
    hipLaunchKernelGGL(create_target_kernel, dim3((w + 255) / 256, (h + TG_ROWS - 1) / TG_ROWS, B), dim3(256), 0, sncal::as_stream(stream),
                       d_kpts, N, sigma, h, w, d_out);
    SNCAL_CHECK_LAUNCH();
    return SNCAL_OK;
}
