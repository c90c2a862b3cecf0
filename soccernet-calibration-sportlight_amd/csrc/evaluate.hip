// Batched camera evaluation (SoccerNet calibration accuracy@t) -- SURVEY 8f N2.
//
// Replaces, per frame, the Python loops of
//   get_polylines               /root/reference/baseline/evaluate_camera.py:14-105
//   distance_to_polyline        evaluate_camera.py:108-157
//   evaluate_camera_prediction  evaluate_camera.py:160-229  (global 2x2 class confusion)
//   mirror_labels + the accuracy choice of evaluate_camera.py:293-320 / src/models/hrnet/metrics.py:97-139
// The sampled pitch model (SoccerPitch.sample_field_points, soccerpitch.py:420-510) is built on the host with numpy
// (its sin/cos must be numpy's, see evaluate.py) and passed in.  One workgroup per frame:
//   1. every thread projects pitch samples with Camera.project_point's arithmetic (fp64, the perspective quotient
//      goes through float32 as baseline/camera.py:247 does);
//   2. one thread per pitch class walks its samples in order and builds the clipped polyline (entering / leaving the
//      image adds the nearest in-image intersection of the crossing segment with the image border lines);
//   3. one thread per (label orientation, class, annotated point) computes its distance to the predicted polyline of
//      that class; a class fails when any of its points is >= threshold away;
//   4. thread 0 assembles both confusions (plain and left/right-mirrored labels), accuracies and the choice.
// fp64 throughout; counts are exact integers, so results equal the oracle's unless a distance sits within rounding
// of the threshold.
#include "common.hpp"
#include "../../include/sncal.h"

namespace {

constexpr int EV_MAX_CLS = 32;

struct V3 { double x, y, z; };
__device__ __forceinline__ V3 cross3(const V3& a, const V3& b) {
    return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}

// nearest in-image intersection of the line through ext (x, y, 1) and prev with the four image border lines
// (evaluate_camera.py:47-69 / :80-98); returns false when none falls inside the image
__device__ bool edge_point(double ex, double ey, const V3& prev, int width, int height, double& ox, double& oy) {
    const V3 ext{ex, ey, 1.0};
    const V3 line = cross3(ext, prev);
    const V3 sides[4] = {{1, 0, 0}, {1, 0, (double)(-width + 1)}, {0, 1, 0}, {0, 1, (double)(-height + 1)}};
    bool found = false;
    double best = 0;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        V3 it = cross3(line, sides[s]);
        it.x /= it.z; it.y /= it.z; it.z /= it.z;            // 0/0 -> NaN -> fails every comparison below
        if (0 <= it.x && it.x < width && 0 <= it.y && it.y < height) {
            const double dx = it.x - ex, dy = it.y - ey, dz = it.z - 1.0;
            const double d = sqrt(dx * dx + dy * dy + dz * dz);
            if (!found || d < best) { found = true; best = d; ox = it.x; oy = it.y; }
        }
    }
    return found;
}

__device__ double point_dist(double px, double py, double qx, double qy) {
    const double dx = px - qx, dy = py - qy;
    return sqrt(dx * dx + dy * dy);
}

__device__ double dist_to_polyline(double px, double py, const double* poly, int n) {
    if (n == 1) return point_dist(px, py, poly[0], poly[1]);
    double best = 0;
    bool nan = false;
    for (int i = 0; i + 1 < n; ++i) {
        const V3 a{poly[2 * i], poly[2 * i + 1], 1.0}, b{poly[2 * i + 2], poly[2 * i + 3], 1.0};
        V3 line = cross3(a, b);
        const double nrm = sqrt(line.x * line.x + line.y * line.y);
        line.x /= nrm; line.y /= nrm; line.z /= nrm;
        const V3 p{px, py, 1.0};
        V3 pr = cross3(cross3(V3{line.x, line.y, 0.0}, p), line);
        pr.x /= pr.z; pr.y /= pr.z; pr.z /= pr.z;
        const double v1x = pr.x - a.x, v1y = pr.y - a.y, v1z = pr.z - a.z;
        const double v2x = b.x - a.x, v2y = b.y - a.y, v2z = b.z - a.z;
        const double k = (v1x * v2x + v1y * v2y + v1z * v2z) / (v2x * v2x + v2y * v2y + v2z * v2z);
        double d;
        if (0 < k && k < 1) {
            const double ex = pr.x - p.x, ey = pr.y - p.y, ez = pr.z - p.z;
            d = sqrt(ex * ex + ey * ey + ez * ez);
        } else {
            d = fmin(point_dist(px, py, a.x, a.y), point_dist(px, py, b.x, b.y));
        }
        if (d != d) nan = true;                                // np.min propagates NaN
        if (i == 0 || d < best) best = d;
    }
    return nan ? NAN : best;
}

__global__ __launch_bounds__(256) void evaluate_kernel(const sncal_camera* __restrict__ cams, int B,
                                                       const double* __restrict__ field, const int* __restrict__ class_start,
                                                       const int* __restrict__ mirror, int n_cls, int n_pts,
                                                       const double* __restrict__ gt, const int* __restrict__ gt_cnt,
                                                       const int* __restrict__ gt_extra, int max_gt, double threshold,
                                                       int width, int height, double ppx, double ppy, float* __restrict__ out,
                                                       double* __restrict__ err, int* __restrict__ cls) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* const ex = reinterpret_cast<double*>(smem);                 // [n_pts] projected x
    double* const ey = ex + n_pts;                                      // [n_pts] projected y
    double* const poly = ey + n_pts;                                    // [2 * n_pts + 2 * n_cls] x 2
    int* const flags = reinterpret_cast<int*>(poly + 2 * (2 * n_pts + 2 * n_cls));     // [n_pts]: bit0 valid, bit1 inside
    __shared__ int poly_n[EV_MAX_CLS], fail[2][EV_MAX_CLS], npts[2][2][EV_MAX_CLS];   // npts[pass][below / not below][class]
    const int b = blockIdx.x, t = threadIdx.x;
    float* const o = out + (size_t)b * 12;
    const sncal_camera cam = cams[b];
    if (cam.status == 0) {                                              // no camera: a "missed" frame (completeness)
        if (t < 12) o[t] = 0.f;
        if (err) for (int i = t; i < 2 * n_cls * max_gt; i += 256) err[(size_t)b * 2 * n_cls * max_gt + i] = NAN;
        if (cls) for (int i = t; i < 2 * n_cls * 4; i += 256) cls[(size_t)b * 2 * n_cls * 4 + i] = 0;
        return;
    }
    for (int i = t; i < n_pts; i += 256) {
        // Camera.project_point, baseline/camera.py:249-268 (zero distortion: the quotient passes through float32)
        const double X = field[3 * i] - cam.position[0], Y = field[3 * i + 1] - cam.position[1], Z = field[3 * i + 2] - cam.position[2];
        const double rx = cam.rotation[0] * X + cam.rotation[1] * Y + cam.rotation[2] * Z;
        const double ry = cam.rotation[3] * X + cam.rotation[4] * Y + cam.rotation[5] * Z;
        const double rz = cam.rotation[6] * X + cam.rotation[7] * Y + cam.rotation[8] * Z;
        int f = 0;
        double x = 0, y = 0;
        if (rz > 1e-3) {                                                 // else project_point returns zeros(3): skipped
            const float dx = (float)(rx / rz), dy = (float)(ry / rz);
            x = (double)dx * cam.fx + ppx;
            y = (double)dy * cam.fy + ppy;
            f = 1 | ((0 <= x && x < width && 0 <= y && y < height) ? 2 : 0);
        }
        ex[i] = x; ey[i] = y; flags[i] = f;
    }
    if (t < EV_MAX_CLS) { poly_n[t] = 0; fail[0][t] = 0; fail[1][t] = 0; npts[0][0][t] = npts[0][1][t] = npts[1][0][t] = npts[1][1][t] = 0; }
    __syncthreads();
    if (t < n_cls) {                                                     // clipped polyline of class t, evaluate_camera.py:41-102
        const int s0 = class_start[t], s1 = class_start[t + 1];
        double* pl = poly + 2 * (2 * s0 + 2 * t);
        int n = 0;
        bool in_img = false;
        V3 prev{0, 0, 0};
        for (int i = s0; i < s1; ++i) {
            const int f = flags[i];
            if (!(f & 1)) continue;
            double qx, qy;
            if (f & 2) {
                if (!in_img && i > s0 && edge_point(ex[i], ey[i], prev, width, height, qx, qy)) { pl[2 * n] = qx; pl[2 * n + 1] = qy; ++n; }
                pl[2 * n] = ex[i]; pl[2 * n + 1] = ey[i]; ++n;
                in_img = true;
            } else if (in_img) {
                if (edge_point(ex[i], ey[i], prev, width, height, qx, qy)) { pl[2 * n] = qx; pl[2 * n + 1] = qy; ++n; }
                in_img = false;
            }
            prev = V3{ex[i], ey[i], 1.0};
        }
        poly_n[t] = n;
    }
    __syncthreads();
    // annotated points vs predicted polylines: pass 0 plain labels, pass 1 mirrored labels (the annotation of class
    // mirror[c] is judged against the prediction of class c)
    const int work = 2 * n_cls * max_gt;
    for (int wi = t; wi < work; wi += 256) {
        const int k = wi % max_gt, c = (wi / max_gt) % n_cls, pass = wi / (max_gt * n_cls);
        const int gc = pass ? mirror[c] : c;
        double d = NAN;
        if (poly_n[c] > 0 && k < gt_cnt[(size_t)b * n_cls + gc]) {
            const double* g = gt + (((size_t)b * n_cls + gc) * max_gt + k) * 2;
            d = dist_to_polyline(g[0], g[1], poly + 2 * (2 * class_start[c] + 2 * c), poly_n[c]);
            const bool below = d < threshold;
            if (!below) atomicOr(&fail[pass][c], 1);
            atomicAdd(&npts[pass][below ? 0 : 1][c], 1);
        }
        if (err) err[(size_t)b * work + wi] = d;                         // dict_errors, evaluate_camera.py:216-219 (NaN = no such point)
    }
    __syncthreads();
    if (cls && t < 2 * n_cls) {                                          // per_class_confusion, evaluate_camera.py:178-214
        const int pass = t / n_cls, c = t - pass * n_cls, gc = pass ? mirror[c] : c;
        const int ann = gt_cnt[(size_t)b * n_cls + gc];
        const bool det = poly_n[c] > 0;
        int* q = cls + (((size_t)b * 2 + pass) * n_cls + c) * 4;
        q[0] = npts[pass][0][c];                                         // [0,0] annotated points within the threshold
        q[1] = npts[pass][1][c];                                         // [0,1] ... beyond it
        q[2] = (!det && ann > 0) ? ann : 0;                              // [1,0] points of a class that was not predicted
        q[3] = (det && ann == 0) ? 1 : 0;                                // predicted, not annotated: the host books 2 (lines) or 9 (circles)
    }
    if (t == 0) {
        float conf[2][4];
        float acc[2];
        for (int pass = 0; pass < 2; ++pass) {
            float tp = 0, fp = 0, fn = (float)gt_extra[b];               // annotated classes the pitch model does not have
            for (int c = 0; c < n_cls; ++c) {
                const int gc = pass ? mirror[c] : c;
                const bool det = poly_n[c] > 0, ann = gt_cnt[(size_t)b * n_cls + gc] > 0;
                if (det && !ann) fp += 1;
                else if (!det && ann) fn += 1;
                else if (det && ann) { if (fail[pass][c]) fp += 1; else tp += 1; }
            }
            conf[pass][0] = tp; conf[pass][1] = fp; conf[pass][2] = fn; conf[pass][3] = 0;
            const float sum = tp + fp + fn;
            acc[pass] = sum > 0 ? tp / sum : 0.f;
        }
        for (int i = 0; i < 4; ++i) { o[i] = conf[0][i]; o[4 + i] = conf[1][i]; }
        o[8] = acc[0]; o[9] = acc[1];
        o[10] = acc[0] > acc[1] ? 1.f : 2.f;                             // evaluate_camera.py:303: plain labels win only if strictly better
        o[11] = 1.f;                                                     // evaluated
    }
}

}  // namespace

extern "C" int sncal_evaluate_cameras_detail(const sncal_camera* d_cams, int B, const double* d_field, const int* d_class_start,
                                      const int* d_mirror, int n_cls, const double* d_gt, const int* d_gt_cnt,
                                      const int* d_gt_extra, int max_gt, double threshold, int img_w, int img_h,
                                      float* d_out, double* d_err, int* d_class_conf, void* stream) {
    SNCAL_CHECK_ARG(B >= 0 && n_cls > 0 && n_cls <= EV_MAX_CLS && max_gt > 0, "sncal_evaluate_cameras: B=%d n_cls=%d max_gt=%d", B, n_cls, max_gt);
    if (B == 0) return SNCAL_OK;
    SNCAL_CHECK_ARG(d_cams && d_field && d_class_start && d_mirror && d_gt && d_gt_cnt && d_gt_extra && d_out, "sncal_evaluate_cameras: null pointer");
    int n_pts = 0;
    SNCAL_CHECK_HIP(hipMemcpyAsync(&n_pts, d_class_start + n_cls, sizeof(int), hipMemcpyDeviceToHost, sncal::as_stream(stream)));
    SNCAL_CHECK_HIP(hipStreamSynchronize(sncal::as_stream(stream)));
    SNCAL_CHECK_ARG(n_pts > 0 && n_pts <= 2048, "sncal_evaluate_cameras: %d pitch samples", n_pts);
    const size_t lds = (size_t)n_pts * 16 + (size_t)(2 * n_pts + 2 * n_cls) * 16 + (size_t)n_pts * 4;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&evaluate_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024);   // + 384 B static
        attr_done = true;
    }
    hipLaunchKernelGGL(evaluate_kernel, dim3(B), dim3(256), lds, sncal::as_stream(stream), d_cams, B, d_field, d_class_start, d_mirror,
                       n_cls, n_pts, d_gt, d_gt_cnt, d_gt_extra, max_gt, threshold, img_w, img_h, img_w / 2.0, img_h / 2.0, d_out, d_err, d_class_conf);
    SNCAL_CHECK_LAUNCH();
    return SNCAL_OK;
}

extern "C" int sncal_evaluate_cameras(const sncal_camera* d_cams, int B, const double* d_field, const int* d_class_start,
                                      const int* d_mirror, int n_cls, const double* d_gt, const int* d_gt_cnt,
                                      const int* d_gt_extra, int max_gt, double threshold, int img_w, int img_h,
                                      float* d_out, void* stream) {
    return sncal_evaluate_cameras_detail(d_cams, B, d_field, d_class_start, d_mirror, n_cls, d_gt, d_gt_cnt, d_gt_extra, max_gt,
                                         threshold, img_w, img_h, d_out, nullptr, nullptr, stream);
}
