// Line peaks -> line equations -> intersection keypoints on the device (bf16/fp32 independent, fp64 arithmetic).
//
// Replaces, per frame, the host-side chain of the reference between the line model and the camera solver:
//   L3  get_line_data / calculate_slope_intercept    /root/reference/src/utils/export_line_result.py:51-131
//   L4  CameraCreator.__init__ lines ingestion       /root/reference/src/models/hrnet/prediction.py:105-124
//       line_eq_intersection                         prediction.py:643-653
//       LINE_CLS order /root/reference/src/datatools/line.py:35-57, LINE_INTERSECTIONS intersections.py:13-44
// (the reference goes through a pickle written by export_line_result.py and read by CameraCreator; here the 30
// candidate keypoints go straight into sncal_calibrate's d_line_pts).
//
// Arithmetic follows the reference's PINNED numpy (1.24.2, value-based scalar promotion): the peak coordinates are
// float32 and stay float32 through `x * scale` and the coordinate differences; the first python-float operand
// (`+ delta`, delta = 1e-5) promotes to float64, and everything after (slope, intercept, intersection) is float64.
// A line needs BOTH peaks with p >= prob_thre; two identical peaks give the reference's (None, None) entry, which
// would raise inside CameraCreator.__init__ -- treated here as "no such line".
#include "common.hpp"

namespace {

// LINE_INTERSECTIONS as LINE_CLS indices: intersection id -> (line1, line2), line1's slope is k1
__constant__ int c_pairs[30][2] = {
    {7, 0}, {7, 14}, {9, 0}, {9, 14}, {18, 8}, {18, 11}, {9, 8}, {9, 11}, {19, 22}, {19, 13},
    {9, 22}, {9, 13}, {9, 4}, {9, 16}, {2, 4}, {2, 16}, {6, 20}, {6, 10}, {12, 20}, {12, 10},
    {21, 15}, {21, 3}, {12, 15}, {12, 3}, {17, 5}, {17, 1}, {12, 5}, {12, 1}, {12, 4}, {12, 16}};

__global__ __launch_bounds__(64) void lines_to_points_kernel(const float* __restrict__ peaks, int B, float scale, double thr,
                                                             float* __restrict__ out) {
    __shared__ double sk[23], sb[23];
    __shared__ int sv[23];
    const int b = blockIdx.x, t = threadIdx.x;
    if (t < 23) {
        const float* p = peaks + ((size_t)b * 23 + t) * 6;
        const float x1 = p[0] * scale, y1 = p[1] * scale, x2 = p[3] * scale, y2 = p[4] * scale;   // float32, as numpy
        bool ok = (double)p[2] >= thr && (double)p[5] >= thr;
        if (x1 == x2 && y1 == y2) ok = false;
        const float dy = y2 - y1, dx = x2 - x1;                                                   // float32 - float32
        const double slope = (double)dy / ((double)dx + 0.00001);
        sk[t] = slope;
        sb[t] = (double)y1 - slope * (double)x1;
        sv[t] = ok ? 1 : 0;
    }
    __syncthreads();
    if (t < 30) {
        const int l1 = c_pairs[t][0], l2 = c_pairs[t][1];
        float x = 0.f, y = 0.f, v = 0.f;
        if (sv[l1] && sv[l2]) {
            const double k1 = sk[l1], b1 = sb[l1], k2 = sk[l2], b2 = sb[l2];
            if (fabs(k1 - k2) > 1e-4) {
                const double xi = (b2 - b1) / (k1 - k2);
                x = (float)xi; y = (float)(k1 * xi + b1); v = 1.f;
            }
        }
        float* o = out + ((size_t)b * 30 + t) * 3;
        o[0] = x; o[1] = y; o[2] = v;
    }
}

}  // namespace

extern "C" int sncal_lines_to_points(const float* d_peaks, int B, float scale, double prob_thre, float* d_out, void* stream) {
    SNCAL_CHECK_ARG(B >= 0, "sncal_lines_to_points: B=%d", B);
    if (B == 0) return SNCAL_OK;
    SNCAL_CHECK_ARG(d_peaks && d_out, "sncal_lines_to_points: null pointer");
    hipLaunchKernelGGL(lines_to_points_kernel, dim3(B), dim3(64), 0, sncal::as_stream(stream), d_peaks, B, scale, prob_thre, d_out);
    SNCAL_CHECK_LAUNCH();
    return SNCAL_OK;
}
