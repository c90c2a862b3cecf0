// Batched camera solve on gfx950: one 64-lane wavefront per frame.
//
// Control flow follows the reference's CameraCreator (/root/reference/src/models/hrnet/prediction.py):
//   __call__ :130-136, iterative_voter :245-257, voter :259-330, original_voter :339-437,
//   get_camera_from_homography :487-520, get_camera_all_points :523-555 (+ quirks Q1/Q2),
//   _reliable/_groundplane/_accurate_points :558-606, get_camera_gen :609-640, good_camera :469-484,
//   opencv_calibration :138-170, opencv_calibration_multiplane :172-243,
// and Camera.solve_pnp / refine_camera / projection_rmse / estimate_calibration_matrix_from_plane_homography
// (/root/reference/baseline/camera.py:92-119, 270-277, 366-426).  The arithmetic behind the cv2 calls
// (findHomography-RANSAC, solvePnPRansac, solvePnPRefineLM, calibrateCamera) is the build's own
// restatement -- specification shared with oracle/solve.py, parity vs OpenCV itself is UNPINNED.
//
// Mapping to the hardware: lane i owns keypoint id i (57 template points <= 64 lanes; a line-intersection
// candidate fills the slot of a missing keypoint with the same id, prediction.py:356-364).  A point subset
// is a 64-bit lane mask.  Everything per point (residuals, Jacobian rows, inlier tests) is lane-parallel;
// normal equations are assembled with butterfly wave reductions (every lane ends with bit-identical sums,
// so all control flow stays wave-uniform); the small dense algebra (8x8 / 6x6 Cholesky, 3x3 adjugates,
// polar iteration) runs redundantly in every lane's registers.  RANSAC hypotheses are lane-parallel too:
// each lane draws its own 4-point sample with a counter-based hash and scores it against all points.
// All arithmetic is fp64.  The solve is latency-bound (~1e5 FLOP per frame): it is reported in frames/s,
// not as a roofline fraction, and runs on its own stream beside the MFMA-bound network.
#include "common.hpp"
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <algorithm>
#include <map>
#include <mutex>
#include <utility>

namespace {

constexpr int NPTS = 57;
constexpr unsigned long long TOP_GATES_MASK = (1ull << 0) | (1ull << 1) | (1ull << 24) | (1ull << 25);
constexpr unsigned long long ALL_MASK = (1ull << NPTS) - 1;
constexpr unsigned long long GROUND_MASK = ALL_MASK & ~TOP_GATES_MASK;
// prediction.py:20-21, 25-26
constexpr unsigned long long GOAL_LEFT_MASK = (1ull << 0) | (1ull << 1) | (1ull << 2) | (1ull << 3) | (1ull << 6) |
                                              (1ull << 7) | (1ull << 10) | (1ull << 11) | (1ull << 12) | (1ull << 13);
constexpr unsigned long long GOAL_RIGHT_MASK = (1ull << 18) | (1ull << 19) | (1ull << 22) | (1ull << 23) | (1ull << 24) |
                                               (1ull << 25) | (1ull << 26) | (1ull << 27) | (1ull << 28) | (1ull << 29);
constexpr unsigned long long KEEP_MASK = ((1ull << 29) - 1) | (1ull << 40) | (1ull << 41) | (1ull << 42) | (1ull << 44) |
                                         (1ull << 45) | (1ull << 48) | (1ull << 51) | (1ull << 52) | (1ull << 55);

__constant__ double c_P64[NPTS * 3];
__constant__ double c_P32[NPTS * 3];
// order of ids inside the goal-plane id lists (for the duplication multiplicity, quirk Q1)
__constant__ int c_goal_left_ids[10] = {0, 1, 2, 3, 6, 7, 10, 11, 12, 13};
__constant__ int c_goal_right_ids[10] = {18, 19, 22, 23, 24, 25, 26, 27, 28, 29};

typedef unsigned long long u64;

// ---- wave helpers ---------------------------------------------------------------------------------
__device__ __forceinline__ double wsum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wmax(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
    return v;
}
// Reductions over aligned groups of W lanes (W = 8, 16, 32, 64) for the packed LM below: every lane ends with the sum / maximum of ITS
// group, and all groups of a wave hold the same points there, so every lane ends with the same bits.  Written with DPP moves -- quad
// swaps, row_half_mirror, row_mirror: three instructions per step and 64-bit value, no LDS round trip -- and, across the four rows of 16,
// v_readlane of the row leaders; as __shfl_xor in a non-inlined device function the same butterflies came out as ds_bpermute pairs
// (708 per iteration at W = 64 where the inlined round-4 code had 352 DPP moves: 2.6 -> 7.0 us per iteration on a 31-point fit, measured).
template <int CTRL>
__device__ __forceinline__ double dpp_mov64(double v) {
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double lane_value64(double v, int lane) {      // wave-uniform copy of lane `lane`'s value
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}
template <int W, bool MAX>
__device__ __forceinline__ double wred_w(double v) {
    auto op = [](double a, double b) { return MAX ? fmax(a, b) : a + b; };
    v = op(v, dpp_mov64<0xB1>(v));                    // quad_perm [1,0,3,2]: lane ^ 1
    v = op(v, dpp_mov64<0x4E>(v));                    // quad_perm [2,3,0,1]: lane ^ 2
    v = op(v, dpp_mov64<0x141>(v));                   // row_half_mirror: the other quad of the 8
    if constexpr (W >= 16) v = op(v, dpp_mov64<0x140>(v));       // row_mirror: the other half of the row of 16
    if constexpr (W == 32) v = op(lane_value64(v, 0), lane_value64(v, 16));          // (both groups hold the same points: group 0's total)
    if constexpr (W == 64) v = op(op(lane_value64(v, 0), lane_value64(v, 16)), op(lane_value64(v, 32), lane_value64(v, 48)));
    return v;
}
template <int W> __device__ __forceinline__ double wsum_w(double v) { return wred_w<W, false>(v); }
template <int W> __device__ __forceinline__ double wmax_w(double v) { return wred_w<W, true>(v); }
__device__ __forceinline__ double bcast(double v, int lane) { return __shfl(v, lane, 64); }
__device__ __forceinline__ int popc64(u64 m) { return __popcll(m); }
__device__ __forceinline__ int kth_set_bit(u64 m, int k) {
    for (int i = 0; i < k; ++i) m &= m - 1;
    return __ffsll((long long)m) - 1;
}

// ---- RANSAC sampler (shared spec: oracle/solve.py::_mix / sample4) -------------------------------------
__device__ __forceinline__ u64 mix64(u64 h, u64 j) {
    u64 z = h * 0x9E3779B97F4A7C15ull + j * 0xBF58476D1CE4E5B9ull + 0x94D049BB133111EBull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__device__ __forceinline__ bool sample4(int h, int n, int (&idx)[4]) {
    int cnt = 0;
    for (int j = 0; j < 16 && cnt < 4; ++j) {
        const int c = (int)((mix64((u64)h, (u64)j) >> 32) % (u64)n);
        bool dup = false;
        for (int k = 0; k < cnt; ++k) dup |= idx[k] == c;
        if (!dup) idx[cnt++] = c;
    }
    return cnt == 4;
}

// ---- small dense algebra (register resident, fully unrolled) -------------------------------------------
__device__ __forceinline__ double det3(const double* m) {
    return m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
}
__device__ __forceinline__ void adj3(const double* m, double* a) {   // adjugate: inv = adj / det
    a[0] = m[4] * m[8] - m[5] * m[7]; a[1] = m[2] * m[7] - m[1] * m[8]; a[2] = m[1] * m[5] - m[2] * m[4];
    a[3] = m[5] * m[6] - m[3] * m[8]; a[4] = m[0] * m[8] - m[2] * m[6]; a[5] = m[2] * m[3] - m[0] * m[5];
    a[6] = m[3] * m[7] - m[4] * m[6]; a[7] = m[1] * m[6] - m[0] * m[7]; a[8] = m[0] * m[4] - m[1] * m[3];
}
__device__ __forceinline__ void mul33(const double* a, const double* b, double* c) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) c[i * 3 + j] = a[i * 3] * b[j] + a[i * 3 + 1] * b[3 + j] + a[i * 3 + 2] * b[6 + j];
}
__device__ __forceinline__ void mul3v(const double* a, const double* v, double* o) {
#pragma unroll
    for (int i = 0; i < 3; ++i) o[i] = a[i * 3] * v[0] + a[i * 3 + 1] * v[1] + a[i * 3 + 2] * v[2];
}

// SPD solve by Cholesky on a packed-full NxN matrix; false when a pivot <= rel_tol * max diag
template <int N>
__device__ __forceinline__ bool chol_solve(const double (&A)[N][N], const double (&b)[N], double (&x)[N]) {
    double L[N][N];
    double dmax = A[0][0];
#pragma unroll
    for (int i = 1; i < N; ++i) dmax = fmax(dmax, A[i][i]);
    if (!(dmax > 0)) return false;
    bool ok = true;
#pragma unroll
    for (int j = 0; j < N; ++j) {
        double d = A[j][j];
#pragma unroll
        for (int k = 0; k < j; ++k) d -= L[j][k] * L[j][k];
        if (!(d > 1e-11 * dmax)) { ok = false; d = 1.0; }
        const double ljj = sqrt(d);
        L[j][j] = ljj;
#pragma unroll
        for (int i = j + 1; i < N; ++i) {
            double s = A[i][j];
#pragma unroll
            for (int k = 0; k < j; ++k) s -= L[i][k] * L[j][k];
            L[i][j] = s / ljj;
        }
    }
    if (!ok) return false;
    double y[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        double s = b[i];
#pragma unroll
        for (int k = 0; k < i; ++k) s -= L[i][k] * y[k];
        y[i] = s / L[i][i];
    }
#pragma unroll
    for (int i = N - 1; i >= 0; --i) {
        double s = y[i];
#pragma unroll
        for (int k = i + 1; k < N; ++k) s -= L[k][i] * x[k];
        x[i] = s / L[i][i];
    }
    return true;
}

__device__ __forceinline__ void polar3(double* R) {   // nearest rotation by Newton iteration
    if (det3(R) < 0) { R[2] = -R[2]; R[5] = -R[5]; R[8] = -R[8]; }
    for (int it = 0; it < 12; ++it) {
        double a[9];
        adj3(R, a);
        const double d = det3(R);
        // inv(R)^T = adj^T / det
        const double n[9] = {a[0] / d, a[3] / d, a[6] / d, a[1] / d, a[4] / d, a[7] / d, a[2] / d, a[5] / d, a[8] / d};
#pragma unroll
        for (int i = 0; i < 9; ++i) R[i] = 0.5 * (R[i] + n[i]);
    }
}

__device__ __forceinline__ void exp_so3(const double* w, double* E) {
    const double th = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
    const double K[9] = {0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0};
    double K2[9];
    mul33(K, K, K2);
    double a, b;
    if (th < 1e-8) { a = 1.0; b = 0.5; }
    else { a = sin(th) / th; b = (1 - cos(th)) / (th * th); }
#pragma unroll
    for (int i = 0; i < 9; ++i) E[i] = (i % 4 == 0 ? 1.0 : 0.0) + a * K[i] + b * K2[i];
}

// ---- homography ----------------------------------------------------------------------------------
__device__ __forceinline__ bool basis_map(const double (&p)[4][2], double* out) {
    const double M[9] = {p[0][0], p[1][0], p[2][0], p[0][1], p[1][1], p[2][1], 1.0, 1.0, 1.0};
    const double det = det3(M);
    double mx = 1.0;
#pragma unroll
    for (int i = 0; i < 4; ++i) mx = fmax(mx, fmax(fabs(p[i][0]), fabs(p[i][1])));
    if (fabs(det) < 1e-9 * mx * mx) return false;
    double a[9];
    adj3(M, a);
    const double rhs[3] = {p[3][0], p[3][1], 1.0};
    double lam[3];
    mul3v(a, rhs, lam);
    lam[0] /= det; lam[1] /= det; lam[2] /= det;
    if (fmin(fabs(lam[0]), fmin(fabs(lam[1]), fabs(lam[2]))) < 1e-9) return false;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) out[i * 3 + j] = M[i * 3 + j] * lam[j];
    return true;
}

__device__ __forceinline__ bool homography_4pt(const double (&s)[4][2], const double (&d)[4][2], double* H) {
    double A[9], B[9];
    if (!basis_map(s, A) || !basis_map(d, B)) return false;
    const double detA = det3(A);
    if (fabs(detA) < 1e-300) return false;
    double adjA[9];
    adj3(A, adjA);
#pragma unroll
    for (int i = 0; i < 9; ++i) adjA[i] /= detA;
    mul33(B, adjA, H);
    if (fabs(H[8]) < 1e-12) return false;
    const double s8 = H[8];
#pragma unroll
    for (int i = 0; i < 9; ++i) H[i] /= s8;
    return true;
}

__device__ __forceinline__ void apply_h(const double* H, double x, double y, double& u, double& v) {
    double w = H[6] * x + H[7] * y + H[8];
    if (fabs(w) < 1e-300) w = 1e-300;
    u = (H[0] * x + H[1] * y + H[2]) / w;
    v = (H[3] * x + H[4] * y + H[5]) / w;
}

// normalised least squares (h33 = 1) + `iters` damped Gauss-Newton steps on the reprojection error
__device__ bool homography_lsq(u64 mask, double sx, double sy, double du, double dv, int iters, double* H) {
    const int lane = threadIdx.x & 63;
    const bool in = (mask >> lane) & 1;
    const double n = (double)popc64(mask);
    const double csx = wsum(in ? sx : 0.0) / n, csy = wsum(in ? sy : 0.0) / n;
    const double cdx = wsum(in ? du : 0.0) / n, cdy = wsum(in ? dv : 0.0) / n;
    const double ms = wsum(in ? sqrt((sx - csx) * (sx - csx) + (sy - csy) * (sy - csy)) : 0.0) / n;
    const double md = wsum(in ? sqrt((du - cdx) * (du - cdx) + (dv - cdy) * (dv - cdy)) : 0.0) / n;
    const double ss = sqrt(2.0) / fmax(ms, 1e-12), sd = sqrt(2.0) / fmax(md, 1e-12);
    const double x = (sx - csx) * ss, y = (sy - csy) * ss, u = (du - cdx) * sd, v = (dv - cdy) * sd;
    double A[8][8], b[8], h[8];
    {
        const double ru[8] = {x, y, 1, 0, 0, 0, -u * x, -u * y};
        const double rv[8] = {0, 0, 0, x, y, 1, -v * x, -v * y};
#pragma unroll
        for (int i = 0; i < 8; ++i) {
#pragma unroll
            for (int j = i; j < 8; ++j) {
                const double s = wsum(in ? ru[i] * ru[j] + rv[i] * rv[j] : 0.0);
                A[i][j] = s; A[j][i] = s;
            }
            b[i] = wsum(in ? ru[i] * u + rv[i] * v : 0.0);
        }
    }
    if (!chol_solve<8>(A, b, h)) return false;
    auto cost = [&](const double* hh) {
        double w = hh[6] * x + hh[7] * y + 1.0;
        if (fabs(w) < 1e-300) w = 1e-300;
        const double pu = (hh[0] * x + hh[1] * y + hh[2]) / w - u, pv = (hh[3] * x + hh[4] * y + hh[5]) / w - v;
        return wsum(in ? pu * pu + pv * pv : 0.0);
    };
    double lam = 1e-3;
    double c0 = cost(h);
    for (int it = 0; it < iters; ++it) {
        const double w = h[6] * x + h[7] * y + 1.0;
        const double pu = (h[0] * x + h[1] * y + h[2]) / w, pv = (h[3] * x + h[4] * y + h[5]) / w;
        const double ju[8] = {x / w, y / w, 1 / w, 0, 0, 0, -pu * x / w, -pu * y / w};
        const double jv[8] = {0, 0, 0, x / w, y / w, 1 / w, -pv * x / w, -pv * y / w};
        const double eu = pu - u, ev = pv - v;
        double g[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
#pragma unroll
            for (int j = i; j < 8; ++j) {
                const double s = wsum(in ? ju[i] * ju[j] + jv[i] * jv[j] : 0.0);
                A[i][j] = s; A[j][i] = s;
            }
            g[i] = -wsum(in ? ju[i] * eu + jv[i] * ev : 0.0);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) A[i][i] += lam * A[i][i];
        double step[8], hn[8];
        double c1 = INFINITY;
        if (chol_solve<8>(A, g, step)) {
#pragma unroll
            for (int i = 0; i < 8; ++i) hn[i] = h[i] + step[i];
            c1 = cost(hn);
        }
        if (c1 < c0) {
#pragma unroll
            for (int i = 0; i < 8; ++i) h[i] = hn[i];
            c0 = c1;
            lam = fmax(lam * 0.1, 1e-12);
        } else {
            lam *= 10.0;
        }
    }
    // H = Td^-1 * Hn * Ts
    const double Hn[9] = {h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7], 1.0};
    const double Ts[9] = {ss, 0, -ss * csx, 0, ss, -ss * csy, 0, 0, 1};
    const double Ti[9] = {1 / sd, 0, cdx, 0, 1 / sd, cdy, 0, 0, 1};
    double t1[9];
    mul33(Hn, Ts, t1);
    mul33(Ti, t1, H);
    if (fabs(H[8]) < 1e-300) return false;
    const double s8 = H[8];
#pragma unroll
    for (int i = 0; i < 9; ++i) H[i] /= s8;
    return true;
}

struct Best { int cnt; double s; int h; };
__device__ __forceinline__ bool better(const Best& a, const Best& b) {   // is a better than b
    return a.cnt > b.cnt || (a.cnt == b.cnt && (a.s < b.s || (a.s == b.s && a.h < b.h)));
}
__device__ __forceinline__ Best wave_best(Best v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        Best q;
        q.cnt = __shfl_xor(v.cnt, o, 64); q.s = __shfl_xor(v.s, o, 64); q.h = __shfl_xor(v.h, o, 64);
        if (better(q, v)) v = q;
    }
    return v;
}

// cv2.findHomography(src, dst, RANSAC, thr) restated (ellipse.py:496-498)
__device__ bool homography_ransac(u64 mask, double sx, double sy, double du, double dv, double thr, double* H) {
    const int lane = threadIdx.x & 63;
    const int n = popc64(mask);
    if (n < 4) return false;
    Best mine{-1, INFINITY, 1 << 30};
    double Hm[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int round = 0; round < 2; ++round) {
        const int h = round * 64 + lane;
        int idx[4] = {0, 0, 0, 0};
        bool ok = sample4(h, n, idx);
        double s[4][2], d[4][2];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int src = kth_set_bit(mask, idx[k]);
            s[k][0] = __shfl(sx, src, 64); s[k][1] = __shfl(sy, src, 64);
            d[k][0] = __shfl(du, src, 64); d[k][1] = __shfl(dv, src, 64);
        }
        double Hh[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};      // never read uninitialised (a failed hypothesis keeps zeros)
        ok = ok && homography_4pt(s, d, Hh);
        int cnt = 0;
        double se = 0;
        for (u64 m = mask; m; m &= m - 1) {
            const int j = __ffsll((long long)m) - 1;
            const double x = bcast(sx, j), y = bcast(sy, j), u = bcast(du, j), v = bcast(dv, j);
            if (ok) {
                double pu, pv;
                apply_h(Hh, x, y, pu, pv);
                const double e2 = (pu - u) * (pu - u) + (pv - v) * (pv - v);
                if (e2 <= thr * thr) { ++cnt; se += e2; }
            }
        }
        const Best cand{ok ? cnt : -1, ok ? se : INFINITY, h};
        const bool take = ok && better(cand, mine);
        mine.cnt = take ? cand.cnt : mine.cnt; mine.s = take ? cand.s : mine.s; mine.h = take ? cand.h : mine.h;
#pragma unroll
        for (int i = 0; i < 9; ++i) Hm[i] = take ? Hh[i] : Hm[i];
    }
    const Best best = wave_best(mine);
    if (best.cnt < 4) return false;
    const int owner = best.h & 63;
    double Hb[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) Hb[i] = bcast(Hm[i], owner);
    double pu, pv;
    apply_h(Hb, sx, sy, pu, pv);
    const double e2 = (pu - du) * (pu - du) + (pv - dv) * (pv - dv);
    const u64 inl = __ballot(((mask >> lane) & 1) && e2 <= thr * thr);
    return homography_lsq(inl, sx, sy, du, dv, 10, H);
}

// camera.py:366-426 in closed form: w = (a,0,a,b,(cy/cx)b,c) spans the null space of the 5x6 system
__device__ bool k_from_homography(const double* H, double cx, double cy, double& fx, double& fy) {
    const double k = cy / cx;
    const double r3[3] = {H[0] * H[1] + H[3] * H[4], (H[0] * H[7] + H[1] * H[6]) + k * (H[3] * H[7] + H[4] * H[6]), H[6] * H[7]};
    const double r4[3] = {(H[0] * H[0] - H[1] * H[1]) + (H[3] * H[3] - H[4] * H[4]),
                          (2 * H[0] * H[6] - 2 * H[1] * H[7]) + k * (2 * H[3] * H[6] - 2 * H[4] * H[7]),
                          H[6] * H[6] - H[7] * H[7]};
    const double a = r3[1] * r4[2] - r3[2] * r4[1], b = r3[2] * r4[0] - r3[0] * r4[2], c = r3[0] * r4[1] - r3[1] * r4[0];
    if (c == 0) return false;
    const double W00 = a / c, W02 = b / c, W12 = k * b / c;
    if (!(W00 > 0)) return false;
    const double L00 = sqrt(W00), L20 = W02 / L00, L21 = W12 / L00;   // W11 == W00
    const double d = 1.0 - L20 * L20 - L21 * L21;
    if (!(d > 0)) return false;
    const double L22 = sqrt(d);
    fx = L22 / L00; fy = L22 / L00;
    return true;
}

// ---- pose ----------------------------------------------------------------------------------------
__device__ bool pose_from_homography(const double* H, double fx, double fy, double cx, double cy, double* R, double* t) {
    const double Ki[9] = {1 / fx, 0, -cx / fx, 0, 1 / fy, -cy / fy, 0, 0, 1};
    double hp[9];
    mul33(Ki, H, hp);
    const double n0 = sqrt(hp[0] * hp[0] + hp[3] * hp[3] + hp[6] * hp[6]);
    const double n1 = sqrt(hp[1] * hp[1] + hp[4] * hp[4] + hp[7] * hp[7]);
    if (n0 < 1e-300 || n1 < 1e-300) return false;
    const double l1 = 1 / n0, l2 = 1 / n1, l3 = sqrt(l1 * l2);
    double r0[3] = {hp[0] * l1, hp[3] * l1, hp[6] * l1}, r1[3] = {hp[1] * l2, hp[4] * l2, hp[7] * l2};
    t[0] = hp[2] * l3; t[1] = hp[5] * l3; t[2] = hp[8] * l3;
    if (t[2] < 0) {
#pragma unroll
        for (int i = 0; i < 3; ++i) { r0[i] = -r0[i]; r1[i] = -r1[i]; t[i] = -t[i]; }
    }
    const double r2[3] = {r0[1] * r1[2] - r0[2] * r1[1], r0[2] * r1[0] - r0[0] * r1[2], r0[0] * r1[1] - r0[1] * r1[0]};
    R[0] = r0[0]; R[1] = r1[0]; R[2] = r2[0];
    R[3] = r0[1]; R[4] = r1[1]; R[5] = r2[1];
    R[6] = r0[2]; R[7] = r1[2]; R[8] = r2[2];
    polar3(R);
    return true;
}

struct K4 { double fx, fy, cx, cy; };

__device__ __forceinline__ void cam_point(const double* R, const double* t, const double* X, double* Xc) {
    Xc[0] = X[0] * R[0] + X[1] * R[1] + X[2] * R[2] + t[0];
    Xc[1] = X[0] * R[3] + X[1] * R[4] + X[2] * R[5] + t[1];
    Xc[2] = X[0] * R[6] + X[1] * R[7] + X[2] * R[8] + t[2];
}
__device__ __forceinline__ double reproj_e2(const double* R, const double* t, const K4& k, const double* X, double u,
                                            double v, double* zout) {
    double Xc[3];
    cam_point(R, t, X, Xc);
    *zout = Xc[2];
    const double zs = fabs(Xc[2]) < 1e-12 ? 1e-12 : Xc[2];
    const double pu = k.fx * Xc[0] / zs + k.cx - u, pv = k.fy * Xc[1] / zs + k.cy - v;
    return pu * pu + pv * pv;
}

// pose rows: residual + Jacobian wrt (w, t) for the left perturbation R <- exp(w) R
__device__ __forceinline__ void pose_rows(const double* R, const double* t, double f_x, double f_y, double cx, double cy,
                                          const double* X, double u, double v, double (&ju)[6], double (&jv)[6],
                                          double& ru, double& rv, double& xn, double& yn) {
    double Xc[3];
    cam_point(R, t, X, Xc);
    const double z = fabs(Xc[2]) < 1e-12 ? 1e-12 : Xc[2];
    const double iz = 1.0 / z;                        // ONE division per point and evaluation (round 5: six, a fifth of an LM iteration's instructions)
    const double x = Xc[0] * iz, y = Xc[1] * iz;
    xn = x; yn = y;
    ru = f_x * x + cx - u; rv = f_y * y + cy - v;
    const double fxz = f_x * iz, fyz = f_y * iz;
    const double du[3] = {fxz, 0.0, -fxz * x}, dv[3] = {0.0, fyz, -fyz * y};
    ju[0] = du[2] * Xc[1] - du[1] * Xc[2]; ju[1] = du[0] * Xc[2] - du[2] * Xc[0]; ju[2] = du[1] * Xc[0] - du[0] * Xc[1];
    ju[3] = du[0]; ju[4] = du[1]; ju[5] = du[2];
    jv[0] = dv[2] * Xc[1] - dv[1] * Xc[2]; jv[1] = dv[0] * Xc[2] - dv[2] * Xc[0]; jv[2] = dv[1] * Xc[0] - dv[0] * Xc[1];
    jv[3] = dv[0]; jv[4] = dv[1]; jv[5] = dv[2];
}

__device__ __forceinline__ void apply_step(const double* R, const double* t, const double* step, double* Rn, double* tn) {
    double E[9];
    exp_so3(step, E);
    mul33(E, R, Rn);
    mul3v(E, t, tn);
    tn[0] += step[3]; tn[1] += step[4]; tn[2] += step[5];
}

// Camera.refine_camera (camera.py:105-119): LM over the pose, K fixed, to convergence
__device__ void refine_pose_lm(u64 mask, double* R, double* t, const K4& k, const double* X, double u, double v,
                               int max_iters, double eps) {
    const int lane = threadIdx.x & 63;
    const bool in = (mask >> lane) & 1;
    auto cost = [&](const double* R_, const double* t_) {
        double z;
        const double e2 = reproj_e2(R_, t_, k, X, u, v, &z);
        return wsum(in ? e2 : 0.0);
    };
    double lam = 1e-3;
    double c0 = cost(R, t);
    for (int it = 0; it < max_iters; ++it) {
        double ju[6], jv[6], ru, rv, xn, yn;
        pose_rows(R, t, k.fx, k.fy, k.cx, k.cy, X, u, v, ju, jv, ru, rv, xn, yn);
        double A[6][6], g[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) {
#pragma unroll
            for (int j = i; j < 6; ++j) {
                const double s = wsum(in ? ju[i] * ju[j] + jv[i] * jv[j] : 0.0);
                A[i][j] = s; A[j][i] = s;
            }
            g[i] = -wsum(in ? ju[i] * ru + jv[i] * rv : 0.0);
        }
        bool improved = false;
        double step[6], dc = 0;
        for (int tr = 0; tr < 12; ++tr) {
            double Ad[6][6];
#pragma unroll
            for (int i = 0; i < 6; ++i)
#pragma unroll
                for (int j = 0; j < 6; ++j) Ad[i][j] = A[i][j] + (i == j ? A[i][i] * lam : 0.0);
            if (!chol_solve<6>(Ad, g, step)) { lam *= 10; continue; }
            double Rn[9], tn[3];
            apply_step(R, t, step, Rn, tn);
            const double c1 = cost(Rn, tn);
            if (c1 < c0) {
#pragma unroll
                for (int i = 0; i < 9; ++i) R[i] = Rn[i];
                t[0] = tn[0]; t[1] = tn[1]; t[2] = tn[2];
                lam = fmax(lam * 0.1, 1e-15);
                dc = c0 - c1; c0 = c1; improved = true;
                break;
            }
            lam *= 10;
        }
        double smax = 0;
#pragma unroll
        for (int i = 0; i < 6; ++i) smax = fmax(smax, fabs(step[i]));
        if (!improved || smax < eps || dc <= 1e-16 * fmax(c0, 1e-30)) break;
    }
    polar3(R);
}

// ---- OpenCV's own minimiser schedules (opencv-python 4.7.0.72, restated from the upstream sources from memory: UNPINNED; shared
// specification with oracle/solve.py lm_solver_pose / cvlevmarq_pose / _joint_cvlevmarq).  Parameters are [rvec, tvec] (Rodrigues), as
// cv.projectPoints differentiates them; SCHED_OPENCV is the default since round 3, SCHED_CONVERGED the build's earlier specification.
constexpr int SCHED_OPENCV = 0, SCHED_CONVERGED = 1;
constexpr double FLT_EPS = 1.1920928955078125e-07, DBL_EPS = 2.220446049250313e-16;

__device__ __forceinline__ void log_so3(const double* R, double* r) {       // cv.Rodrigues(matrix -> vector), R orthonormal
    const double c = fmin(1.0, fmax(-1.0, (R[0] + R[4] + R[8] - 1.0) * 0.5));
    const double th = acos(c);
    const double a[3] = {(R[7] - R[5]) * 0.5, (R[2] - R[6]) * 0.5, (R[3] - R[1]) * 0.5};      // sin(th) * axis
    const double sn = sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
    if (sn < 1e-5) {
        if (c > 0) { r[0] = a[0]; r[1] = a[1]; r[2] = a[2]; return; }
        // th ~ pi: axis from the symmetric part (R + I) / 2 = axis axis^T, sign from what is left of the antisymmetric part
        const double B[9] = {(R[0] + 1) * 0.5, R[1] * 0.5, R[2] * 0.5, R[3] * 0.5, (R[4] + 1) * 0.5, R[5] * 0.5, R[6] * 0.5, R[7] * 0.5, (R[8] + 1) * 0.5};
        const double d0 = sqrt(fmax(B[0], 0.0)), d1 = sqrt(fmax(B[4], 0.0)), d2 = sqrt(fmax(B[8], 0.0));
        const int kx = d0 >= d1 && d0 >= d2 ? 0 : (d1 >= d2 ? 1 : 2);
        const double dk = fmax(kx == 0 ? d0 : kx == 1 ? d1 : d2, 1e-300);
        double ax[3] = {B[kx] / dk, B[3 + kx] / dk, B[6 + kx] / dk};
        const double n = fmax(sqrt(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]), 1e-300);
        const double sg = (a[0] * ax[0] + a[1] * ax[1] + a[2] * ax[2]) < 0 ? -1.0 : 1.0;
        r[0] = sg * ax[0] / n * th; r[1] = sg * ax[1] / n * th; r[2] = sg * ax[2] / n * th;
        return;
    }
    const double q = th / sn;
    r[0] = a[0] * q; r[1] = a[1] * q; r[2] = a[2] * q;
}

// The two maps of a rotation vector with the angle and its sine / cosine given (ONE sincos for both, pose_normal_eq): the same
// formulas as exp_so3 / left_jacobian_so3
struct RotAngle { double th, sn, cs; };
__device__ __forceinline__ RotAngle rot_angle(const double* w) {
    RotAngle a;
    a.th = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
    sincos(a.th, &a.sn, &a.cs);
    return a;
}
__device__ __forceinline__ void exp_so3_a(const double* w, const RotAngle& q, double* E) {
    const double K[9] = {0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0};
    double K2[9];
    mul33(K, K, K2);
    double a, b;
    if (q.th < 1e-8) { a = 1.0; b = 0.5; }
    else { a = q.sn / q.th; b = (1 - q.cs) / (q.th * q.th); }
#pragma unroll
    for (int i = 0; i < 9; ++i) E[i] = (i % 4 == 0 ? 1.0 : 0.0) + a * K[i] + b * K2[i];
}
__device__ __forceinline__ void left_jacobian_so3_a(const double* w, const RotAngle& q, double* J) {
    const double K[9] = {0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0};
    double K2[9];
    mul33(K, K, K2);
    double a, b;
    if (q.th < 1e-6) { a = 0.5; b = 1.0 / 6.0; }
    else { a = (1 - q.cs) / (q.th * q.th); b = (q.th - q.sn) / (q.th * q.th * q.th); }
#pragma unroll
    for (int i = 0; i < 9; ++i) J[i] = (i % 4 == 0 ? 1.0 : 0.0) + a * K[i] + b * K2[i];
}

// exp(r + d) ~ exp(J_l(r) d) exp(r)
__device__ __forceinline__ void left_jacobian_so3(const double* w, double* J) {
    const double th = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
    const double K[9] = {0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0};
    double K2[9];
    mul33(K, K, K2);
    double a, b;
    if (th < 1e-6) { a = 0.5; b = 1.0 / 6.0; }
    else { a = (1 - cos(th)) / (th * th); b = (th - sin(th)) / (th * th * th); }
#pragma unroll
    for (int i = 0; i < 9; ++i) J[i] = (i % 4 == 0 ? 1.0 : 0.0) + a * K[i] + b * K2[i];
}

// residual + Jacobian wrt (rvec, tvec): R = exp(rvec), Jl = left_jacobian_so3(rvec) computed by the caller
__device__ __forceinline__ void pose_rows_rvec(const double* R, const double* Jl, const double* t, double f_x, double f_y, double cx,
                                               double cy, const double* X, double u, double v, double (&ju)[6], double (&jv)[6],
                                               double& ru, double& rv, double& xn, double& yn) {
    double Xr[3];
    const double zero[3] = {0, 0, 0};
    cam_point(R, zero, X, Xr);
    const double Xc[3] = {Xr[0] + t[0], Xr[1] + t[1], Xr[2] + t[2]};
    const double z = fabs(Xc[2]) < 1e-12 ? 1e-12 : Xc[2];
    const double x = Xc[0] / z, y = Xc[1] / z;
    xn = x; yn = y;
    ru = f_x * x + cx - u; rv = f_y * y + cy - v;
    const double du[3] = {f_x / z, 0.0, -f_x * x / z}, dv[3] = {0.0, f_y / z, -f_y * y / z};
    // d . (w x Xr) = w . (Xr x d): a rotation about the camera origin moves the ROTATED point only (tvec is its own parameter)
    const double wu[3] = {Xr[1] * du[2] - Xr[2] * du[1], Xr[2] * du[0] - Xr[0] * du[2], Xr[0] * du[1] - Xr[1] * du[0]};
    const double wv[3] = {Xr[1] * dv[2] - Xr[2] * dv[1], Xr[2] * dv[0] - Xr[0] * dv[2], Xr[0] * dv[1] - Xr[1] * dv[0]};
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        ju[j] = wu[0] * Jl[j] + wu[1] * Jl[3 + j] + wu[2] * Jl[6 + j];
        jv[j] = wv[0] * Jl[j] + wv[1] * Jl[3 + j] + wv[2] * Jl[6 + j];
        ju[3 + j] = du[j]; jv[3 + j] = dv[j];
    }
}

// cv::solve(A, b, DECOMP_EIG / DECOMP_SVD) for a symmetric 6x6 system: Cholesky when A is positive definite, else the minimum-norm
// solution from a cyclic Jacobi eigen-decomposition with eigenvalues below 2 eps sum|w| dropped
// (The rotation indices are compile-time constants -- fully unrolled pair loop -- so that M and V live in registers: with run-time
// indices they sat in scratch memory and one fallback solve cost tens of thousands of clocks; the slow LM runs of degenerate
// candidates are exactly the ones that take this path every iteration.  Same operations in the same order as before.)
struct Sym6 { bool chol; double L[6][6]; double M[6][6], V[6][6]; double thr; };
__device__ __forceinline__ void sym_factor6(const double (&A)[6][6], Sym6& F) {
    // Cholesky factor (chol_solve's): usable when every pivot passes
    double dmax = A[0][0];
#pragma unroll
    for (int i = 1; i < 6; ++i) dmax = fmax(dmax, A[i][i]);
    bool ok = dmax > 0;
    if (ok) {
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            double d = A[j][j];
#pragma unroll
            for (int k = 0; k < j; ++k) d -= F.L[j][k] * F.L[j][k];
            if (!(d > 1e-11 * dmax)) { ok = false; d = 1.0; }
            const double ljj = sqrt(d);
            F.L[j][j] = ljj;
#pragma unroll
            for (int i = j + 1; i < 6; ++i) {
                double s = A[i][j];
#pragma unroll
                for (int k = 0; k < j; ++k) s -= F.L[i][k] * F.L[j][k];
                F.L[i][j] = s / ljj;
            }
        }
    }
    F.chol = ok;
    if (ok) return;
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) { F.M[i][j] = A[i][j]; F.V[i][j] = i == j ? 1.0 : 0.0; }
    for (int sweep = 0; sweep < 12; ++sweep) {
        double off = 0;
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int j = i + 1; j < 6; ++j) off += F.M[i][j] * F.M[i][j];
        if (!(off > 1e-300)) break;
#pragma unroll
        for (int p_ = 0; p_ < 5; ++p_)
#pragma unroll
            for (int q_ = p_ + 1; q_ < 6; ++q_) {
                const double apq = F.M[p_][q_];
                if (apq != 0.0) {
                    const double th = (F.M[q_][q_] - F.M[p_][p_]) / (2.0 * apq);
                    const double tt = (th >= 0 ? 1.0 : -1.0) / (fabs(th) + sqrt(th * th + 1.0));
                    const double cs = 1.0 / sqrt(tt * tt + 1.0), sn = tt * cs;
#pragma unroll
                    for (int k2 = 0; k2 < 6; ++k2) { const double a = F.M[k2][p_], bb = F.M[k2][q_]; F.M[k2][p_] = cs * a - sn * bb; F.M[k2][q_] = sn * a + cs * bb; }
#pragma unroll
                    for (int k2 = 0; k2 < 6; ++k2) { const double a = F.M[p_][k2], bb = F.M[q_][k2]; F.M[p_][k2] = cs * a - sn * bb; F.M[q_][k2] = sn * a + cs * bb; }
#pragma unroll
                    for (int k2 = 0; k2 < 6; ++k2) { const double a = F.V[k2][p_], bb = F.V[k2][q_]; F.V[k2][p_] = cs * a - sn * bb; F.V[k2][q_] = sn * a + cs * bb; }
                }
            }
    }
    double sw = 0;
#pragma unroll
    for (int i = 0; i < 6; ++i) sw += fabs(F.M[i][i]);
    F.thr = 2.0 * DBL_EPS * sw;
}
__device__ __forceinline__ void sym_apply6(const Sym6& F, const double (&b)[6], double (&x)[6]) {
    if (F.chol) {
        double y[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            double s = b[i];
#pragma unroll
            for (int k = 0; k < i; ++k) s -= F.L[i][k] * y[k];
            y[i] = s / F.L[i][i];
        }
#pragma unroll
        for (int i = 5; i >= 0; --i) {
            double s = y[i];
#pragma unroll
            for (int k = i + 1; k < 6; ++k) s -= F.L[k][i] * x[k];
            x[i] = s / F.L[i][i];
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) x[i] = 0;
#pragma unroll
    for (int e = 0; e < 6; ++e) {
        if (!(fabs(F.M[e][e]) > F.thr)) continue;
        double pj = 0;
#pragma unroll
        for (int i = 0; i < 6; ++i) pj += F.V[i][e] * b[i];
        pj /= F.M[e][e];
#pragma unroll
        for (int i = 0; i < 6; ++i) x[i] += F.V[i][e] * pj;
    }
}
__device__ void sym_solve6(const double (&A)[6][6], const double (&b)[6], double (&x)[6]) {
    Sym6 F;
    sym_factor6(A, F);
    sym_apply6(F, b, x);
}

// normal equations of the pose problem at x = [rvec, tvec]: A = J^T J, g = J^T r, S = |r|^2, rinf = |r|_inf (all wave-uniform)
// `rc` (optional): the rotation of x -- angle, sine, cosine, matrix.  A call with want_j = false FILLS it; a call with want_j = true and
// rc->valid USES it instead of recomputing: lm_solver_pose linearises an accepted step at exactly the point it has just evaluated, and
// the sine / cosine / matrix of the rotation vector were 40 % of that evaluation's clocks (SNCAL_LM_TIMING: 5.3k clk per Jacobian
// evaluation, 2.2k per trial on the 8-point fit).
struct RotCache { RotAngle q; double R[9]; bool valid; };
template <int W = 64>
__device__ void pose_normal_eq(u64 mask, const double* x, const K4& k, const double* X, double u, double v, bool want_j,
                               double (&A)[6][6], double (&g)[6], double& S, double& rinf, RotCache* rc = nullptr) {
    const int lane = threadIdx.x & 63;
    const bool in = (mask >> lane) & 1;
    double R[9], Jl[9];
    RotAngle q;
    if (rc != nullptr && want_j && rc->valid) {
        q = rc->q;
#pragma unroll
        for (int i = 0; i < 9; ++i) R[i] = rc->R[i];
    } else {
        q = rot_angle(x);
        exp_so3_a(x, q, R);
        if (rc != nullptr) {
            rc->q = q; rc->valid = true;
#pragma unroll
            for (int i = 0; i < 9; ++i) rc->R[i] = R[i];
        }
    }
    double ju[6], jv[6], ru, rv, xn, yn;
    if (want_j) {
        left_jacobian_so3_a(x, q, Jl);
        pose_rows_rvec(R, Jl, x + 3, k.fx, k.fy, k.cx, k.cy, X, u, v, ju, jv, ru, rv, xn, yn);
#pragma unroll
        for (int i = 0; i < 6; ++i) {
#pragma unroll
            for (int j = i; j < 6; ++j) {
                const double s = wsum_w<W>(in ? ju[i] * ju[j] + jv[i] * jv[j] : 0.0);
                A[i][j] = s; A[j][i] = s;
            }
            g[i] = wsum_w<W>(in ? ju[i] * ru + jv[i] * rv : 0.0);
        }
    } else {
        double Xc[3];
        cam_point(R, x + 3, X, Xc);
        const double z = fabs(Xc[2]) < 1e-12 ? 1e-12 : Xc[2];
        const double iz = 1.0 / z;                    // (the same x = X / z, y = Y / z as pose_rows_rvec: a step is judged on the residual it will be linearised at)
        ru = k.fx * (Xc[0] * iz) + k.cx - u; rv = k.fy * (Xc[1] * iz) + k.cy - v;
    }
    S = wsum_w<W>(in ? ru * ru + rv * rv : 0.0);
    rinf = wmax_w<W>(in ? fmax(fabs(ru), fabs(rv)) : 0.0);
}

// Cholesky of a 6 x 6 system for the LM below, with the diagonal kept as RECIPROCALS: L[i][j] = s * rinv[j] and the substitutions
// multiply -- 6 reciprocal square roots per factorisation and no division in a solve, where chol_solve's form has 6 square roots + 15
// divisions per factorisation and 12 dependent divisions per solve.  refine_camera's slow fits (20000 iterations at the reference's
// criterion, camera.py:116) are ONE wavefront issuing ~2200 dependent fp64 instructions per iteration: 7.8 us each, 160 ms for the fit that
// sets the latency of a batch's solve (NOTES/design_history_r1_r5.md §11.1); a fifth of those instructions were divisions.  Same pivot rule as sym_factor6 (a
// failing pivot sends the caller to its eigen-decomposition fallback); results differ from the dividing form by rounding only.
struct Chol6 { double L[6][6]; double rinv[6]; bool ok; };
__device__ __forceinline__ void chol6_factor(const double (&A)[6][6], Chol6& F) {
    double dmax = A[0][0];
#pragma unroll
    for (int i = 1; i < 6; ++i) dmax = fmax(dmax, A[i][i]);
    bool ok = dmax > 0;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        double d = A[j][j];
#pragma unroll
        for (int k = 0; k < j; ++k) d -= F.L[j][k] * F.L[j][k];
        if (!(d > 1e-11 * dmax)) { ok = false; d = 1.0; }
        const double ri = rsqrt(d);
        F.rinv[j] = ri;
#pragma unroll
        for (int i = j + 1; i < 6; ++i) {
            double sacc = A[i][j];
#pragma unroll
            for (int k = 0; k < j; ++k) sacc -= F.L[i][k] * F.L[j][k];
            F.L[i][j] = sacc * ri;
        }
    }
    F.ok = ok;
}
__device__ __forceinline__ void chol6_apply(const Chol6& F, const double (&b)[6], double (&x)[6]) {
    double y[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        double sacc = b[i];
#pragma unroll
        for (int k = 0; k < i; ++k) sacc -= F.L[i][k] * y[k];
        y[i] = sacc * F.rinv[i];
    }
#pragma unroll
    for (int i = 5; i >= 0; --i) {
        double sacc = y[i];
#pragma unroll
        for (int k = i + 1; k < 6; ++k) sacc -= F.L[k][i] * x[k];
        x[i] = sacc * F.rinv[i];
    }
}
// max_e |(A^-1)_ee| from the factor: A^-1 = L^-T L^-1, so (A^-1)_ee = sum_i (L^-1)_ie^2 -- column e of L^-1 by one forward
// substitution of a unit vector, instead of six full solves
__device__ __forceinline__ double chol6_inv_diag_max(const Chol6& F) {
    double mx = 0;
#pragma unroll
    for (int e = 0; e < 6; ++e) {
        double z[6], acc = 0;
#pragma unroll
        for (int i = e; i < 6; ++i) {
            double sacc = i == e ? 1.0 : 0.0;
#pragma unroll
            for (int k = e; k < i; ++k) sacc -= F.L[i][k] * z[k];
            z[i] = sacc * F.rinv[i];
            acc += z[i] * z[i];
        }
        mx = fmax(mx, acc);
    }
    return mx;
}

// cv.solvePnPRefineLM = LMSolver::run (calib3d levmarq.cpp): D = diag(J^T J) fixed at the start, lambda_0 = 1, gain-ratio schedule
// (0.25 / 0.75, nu in [2, 10], lambda -> 0 below lambda_c), accept when the error falls, stop on |d|_inf < eps or |r|_inf < eps.
// W: the points sit in every aligned group of W lanes (cam_refine packs them when there are few: a reduction is then log2 W butterfly
// steps instead of six); W = 64 is the plain one-point-per-lane layout.
// (One copy of the loop per W in a kernel, not one per call site: a real function -- with every operand passed BY VALUE, in registers;
// through pointers the pose and the lane's point would live in scratch memory and every iteration would fetch them from there: measured,
// 2.6 -> 7.0 us per iteration on a 31-point fit.)
struct LmPose { double R[9], t[3]; };
template <int W>
__device__ __attribute__((noinline)) LmPose lm_solver_pose_fn(u64 mask, LmPose io, K4 k, double X0, double X1, double X2, double u, double v, int max_iters, double eps) {
    double R[9], t[3];
#pragma unroll
    for (int i = 0; i < 9; ++i) R[i] = io.R[i];
    t[0] = io.t[0]; t[1] = io.t[1]; t[2] = io.t[2];
    const double X[3] = {X0, X1, X2};
    double x[6];
    log_so3(R, x);
    x[3] = t[0]; x[4] = t[1]; x[5] = t[2];
    double A[6][6], g[6], S, rinf;
    pose_normal_eq<W>(mask, x, k, X, u, v, true, A, g, S, rinf);
    double D[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) D[i] = A[i][i];
    double lam = 1.0, lc = 0.75;
#ifdef SNCAL_LM_TIMING
    unsigned long long tq[5] = {0, 0, 0, 0, 0}, tp = __builtin_amdgcn_s_memtime();
#define LM_LAP(k) do { const unsigned long long tn_ = __builtin_amdgcn_s_memtime(); tq[k] += tn_ - tp; tp = tn_; } while (0)
#else
#define LM_LAP(k) do {} while (0)
#endif
    for (int it = 0;;) {
        double Ap[6][6], d[6], xd[6];
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int j = 0; j < 6; ++j) Ap[i][j] = A[i][j] + (i == j ? lam * D[i] : 0.0);
        {
            Chol6 F;
            chol6_factor(Ap, F);
            if (F.ok) chol6_apply(F, g, d);
            else sym_solve6(Ap, g, d);                // not positive definite: the eigen-decomposition fallback (cv::solve DECOMP_EIG)
        }
        LM_LAP(0);
#pragma unroll
        for (int i = 0; i < 6; ++i) xd[i] = x[i] - d[i];
        // (Measured and not kept: the trial evaluated WITH its normal equations, so that an accepted step is not evaluated twice -- same bits,
        // but 42 more live doubles pushed every width of this function into scratch memory: 2.97 -> 3.43 us per iteration on the 8-point
        // crawl, 3.0 -> 4.6 on a 31-point fit.)
        double A2[6][6], g2[6], Sd, rinf_d;
        RotCache rc;
        rc.valid = false;
        pose_normal_eq<W>(mask, xd, k, X, u, v, false, A2, g2, Sd, rinf_d, &rc);
        LM_LAP(1);
        double dS = 0, dv_ = 0, dmax = 0;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            double Ad = 0;
#pragma unroll
            for (int j = 0; j < 6; ++j) Ad += A[i][j] * d[j];
            dS += d[i] * (2.0 * g[i] - Ad);
            dv_ += d[i] * g[i];
            dmax = fmax(dmax, fabs(d[i]));
        }
        const double Rg = (S - Sd) / (fabs(dS) > DBL_EPS ? dS : 1.0);
        if (Rg > 0.75) {
            lam *= 0.5;
            if (lam < lc) lam = 0.0;
        } else if (Rg < 0.25) {
            double nu = (Sd - S) / (fabs(dv_) > DBL_EPS ? dv_ : 1.0) + 2.0;
            nu = fmin(fmax(nu, 2.0), 10.0);
            if (lam == 0.0) {
                double mx = DBL_EPS;
                // (lambda = 0 means the step's matrix WAS A, and keeping its factor for here would save this factorisation -- measured: no
                // gain, and the longer-lived factor pushed the function into scratch memory: 3.1 -> 4.3 us per iteration on a 31-point fit)
                Chol6 C;
                chol6_factor(A, C);
                if (C.ok) {
                    mx = fmax(mx, chol6_inv_diag_max(C));
                } else {
                    Sym6 F;                          // one factorisation for the six columns of the inverse
                    sym_factor6(A, F);
#pragma unroll
                    for (int e = 0; e < 6; ++e) {
                        double unit[6] = {0, 0, 0, 0, 0, 0}, col[6];
                        unit[e] = 1.0;
                        sym_apply6(F, unit, col);
                        mx = fmax(mx, fabs(col[e]));
                    }
                }
                lam = lc = 1.0 / mx;
                nu *= 0.5;
            }
            lam *= nu;
        }
        LM_LAP(2);
        if (Sd < S) {
#pragma unroll
            for (int i = 0; i < 6; ++i) x[i] = xd[i];
            pose_normal_eq<W>(mask, x, k, X, u, v, true, A, g, S, rinf, &rc);      // (x = xd: the rotation the trial has just built)
        }
        LM_LAP(3);
        ++it;
        if (!(it < max_iters && dmax >= eps && rinf >= eps)) {
#ifdef SNCAL_LM_TIMING
            if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) printf("LM W=%d its %d: clocks per iteration: solve %.0f trial %.0f gain+inverse %.0f accept+J %.0f\n", W, it, (double)tq[0] / it, (double)tq[1] / it, (double)tq[2] / it, (double)tq[3] / it);
#endif
            break;
        }
    }
    exp_so3(x, R);
    LmPose out;
#pragma unroll
    for (int i = 0; i < 9; ++i) out.R[i] = R[i];
    out.t[0] = x[3]; out.t[1] = x[4]; out.t[2] = x[5];
    return out;
}
template <int W = 64>
__device__ __forceinline__ void lm_solver_pose(u64 mask, double* R, double* t, const K4& k, const double* X, double u, double v, int max_iters, double eps) {
    LmPose io;
#pragma unroll
    for (int i = 0; i < 9; ++i) io.R[i] = R[i];
    io.t[0] = t[0]; io.t[1] = t[1]; io.t[2] = t[2];
    const LmPose o = lm_solver_pose_fn<W>(mask, io, k, X[0], X[1], X[2], u, v, max_iters, eps);
#pragma unroll
    for (int i = 0; i < 9; ++i) R[i] = o.R[i];
    t[0] = o.t[0]; t[1] = o.t[1]; t[2] = o.t[2];
}

// lm_solver_pose with the points PACKED when there are few: point r (the r-th set bit of `mask`) goes to lane g * W + r of every aligned
// group g of W = 8 / 16 / 32 lanes, so that the 31 reductions of an iteration are 3 / 4 / 5 butterfly steps instead of 6 and every lane still
// ends with the same sums (all groups hold the same points).  The slow fits are the ill-posed ones, and those have few points (the
// bench's frame 16: 8).  More than 32 points: the plain layout.  The summation order depends on W, the results on nothing else.
template <int W>
__device__ __forceinline__ void lm_solver_pose_packed(u64 mask, int n, double* R, double* t, const K4& k, const double* X, double u, double v,
                                                      int max_iters, double eps) {
    const int lane = threadIdx.x & 63, r = lane & (W - 1);
    u64 m = mask;
    for (int i = 0; i < r; ++i) m &= m - 1;                      // (per-lane trip count, once per fit)
    const int src = (r < n && m) ? __ffsll((long long)m) - 1 : lane;
    const double Xp[3] = {__shfl(X[0], src, 64), __shfl(X[1], src, 64), __shfl(X[2], src, 64)};
    const double up = __shfl(u, src, 64), vp = __shfl(v, src, 64);
    const u64 grp = n >= 64 ? ~0ull : ((1ull << n) - 1);
    u64 pm = 0;
#pragma unroll
    for (int g = 0; g < 64 / W; ++g) pm |= grp << (g * W);
    lm_solver_pose<W>(pm, R, t, k, Xp, up, vp, max_iters, eps);
}
__device__ void lm_solver_pose_auto(u64 mask, double* R, double* t, const K4& k, const double* X, double u, double v, int max_iters, double eps) {
    const int n = popc64(mask);
    if (n >= 1 && n <= 8) lm_solver_pose_packed<8>(mask, n, R, t, k, X, u, v, max_iters, eps);
    else if (n <= 16 && n >= 1) lm_solver_pose_packed<16>(mask, n, R, t, k, X, u, v, max_iters, eps);
    else if (n <= 32 && n >= 1) lm_solver_pose_packed<32>(mask, n, R, t, k, X, u, v, max_iters, eps);
    else lm_solver_pose<64>(mask, R, t, k, X, u, v, max_iters, eps);
}

// cvFindExtrinsicCameraParams2's refinement (solvePnPRansac's final SOLVEPNP_ITERATIVE refit, calibrateCamera's per-view initial
// extrinsics): CvLevMarq over [rvec, tvec] -- lambda = 10^k, k_0 = -3, diagonal x (1 + lambda), steps from the same normal equations
// until the error no longer grows (k + 1 per rejection, up to 16), an accepted step lowers k; criteria (max_iter, eps on |dx| / |x|)
__device__ void cvlevmarq_pose(u64 mask, double* R, double* t, const K4& k, const double* X, double u, double v, int max_iter, double eps) {
    double x[6];
    log_so3(R, x);
    x[3] = t[0]; x[4] = t[1]; x[5] = t[2];
    double A[6][6], g[6], e_prev, rinf;
    pose_normal_eq(mask, x, k, X, u, v, true, A, g, e_prev, rinf);
    int kk = -3, iters = 0;
    for (;;) {
        double cand[6], e = INFINITY;
        bool have = false;
        for (;;) {
            const double lam = pow(10.0, (double)kk);
            double Ad[6][6], d[6];
#pragma unroll
            for (int i = 0; i < 6; ++i)
#pragma unroll
                for (int j = 0; j < 6; ++j) Ad[i][j] = A[i][j] + (i == j ? lam * A[i][i] : 0.0);
            sym_solve6(Ad, g, d);                               // cv::solve(..., DECOMP_SVD): a step even when not positive definite
            {
#pragma unroll
                for (int i = 0; i < 6; ++i) cand[i] = x[i] - d[i];
                double A2[6][6], g2[6], r2;
                pose_normal_eq(mask, cand, k, X, u, v, false, A2, g2, e, r2);
                have = true;
            }
            if (!(e > e_prev)) break;
            if (++kk > 16) break;
        }
        if (!have || !isfinite(e)) break;                    // no usable step at any damping: keep the last parameters
        kk = max(kk - 1, -16);
        double dn = 0, pn = 0;
#pragma unroll
        for (int i = 0; i < 6; ++i) { dn += (cand[i] - x[i]) * (cand[i] - x[i]); pn += x[i] * x[i]; x[i] = cand[i]; }
        ++iters;
        if (iters >= max_iter || sqrt(dn) / fmax(sqrt(pn), 1e-300) < eps) break;
        pose_normal_eq(mask, x, k, X, u, v, true, A, g, e_prev, rinf);
    }
    exp_so3(x, R);
    t[0] = x[3]; t[1] = x[4]; t[2] = x[5];
}

__device__ __forceinline__ void refit_pose(int sched, u64 mask, double* R, double* t, const K4& k, const double* X, double u, double v) {
    if (sched == SCHED_OPENCV) cvlevmarq_pose(mask, R, t, k, X, u, v, 20, FLT_EPS);
    else refine_pose_lm(mask, R, t, k, X, u, v, 20, 1e-10);
}

// Lane-local damped Gauss-Newton polish of a minimal-sample pose on its own 4 (z=0) points.  The closed-form
// homography decomposition is badly conditioned for long focal lengths; a few iterations repair it.
__device__ void polish4(double* R, double* t, const K4& k, const double (&s)[4][2], const double (&d)[4][2]) {
    auto cost = [&](const double* R_, const double* t_) {
        double c = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const double Xq[3] = {s[q][0], s[q][1], 0.0};
            double z;
            c += reproj_e2(R_, t_, k, Xq, d[q][0], d[q][1], &z);
        }
        return c;
    };
    double c0 = cost(R, t);
    for (int it = 0; it < 8; ++it) {
        double A[6][6], g[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) { g[i] = 0; for (int j = 0; j < 6; ++j) A[i][j] = 0; }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const double Xq[3] = {s[q][0], s[q][1], 0.0};
            double ju[6], jv[6], ru, rv, xn, yn;
            pose_rows(R, t, k.fx, k.fy, k.cx, k.cy, Xq, d[q][0], d[q][1], ju, jv, ru, rv, xn, yn);
#pragma unroll
            for (int i = 0; i < 6; ++i) {
#pragma unroll
                for (int j = 0; j < 6; ++j) A[i][j] += ju[i] * ju[j] + jv[i] * jv[j];
                g[i] -= ju[i] * ru + jv[i] * rv;
            }
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) A[i][i] += 1e-3 * A[i][i];
        double step[6] = {0, 0, 0, 0, 0, 0}, Rn[9], tn[3];
        if (!chol_solve<6>(A, g, step)) break;
        apply_step(R, t, step, Rn, tn);
        const double c1 = cost(Rn, tn);
        if (!(c1 < c0)) break;
#pragma unroll
        for (int i = 0; i < 9; ++i) R[i] = Rn[i];
        t[0] = tn[0]; t[1] = tn[1]; t[2] = tn[2];
        c0 = c1;
    }
}

// Camera.solve_pnp (camera.py:92-103): planar minimal solver on the z=0 points (64 lane-parallel 4-point
// hypotheses + one least-squares homography over all of them), 8 px inliers, LM refit on the inliers
__device__ bool pnp_ransac(int sched, u64 mask, u64 gmask, const K4& k, const double* X, double u, double v, double* R, double* t) {
    const int lane = threadIdx.x & 63;
    const int n = popc64(gmask);
    if (n < 4) return false;
    int idx[4] = {0, 0, 0, 0};
    bool ok = sample4(lane, n, idx);
    double s[4][2], d[4][2];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int src = kth_set_bit(gmask, idx[q]);
        s[q][0] = __shfl(X[0], src, 64); s[q][1] = __shfl(X[1], src, 64);
        d[q][0] = __shfl(u, src, 64); d[q][1] = __shfl(v, src, 64);
    }
    double Hh[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, Rh[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, th[3] = {0, 0, 1};
    ok = ok && homography_4pt(s, d, Hh);
    ok = ok && pose_from_homography(Hh, k.fx, k.fy, k.cx, k.cy, Rh, th);
    if (ok) polish4(Rh, th, k, s, d);
    int cnt = 0;
    double se = 0;
    for (u64 m = mask; m; m &= m - 1) {
        const int j = __ffsll((long long)m) - 1;
        const double Xj[3] = {bcast(X[0], j), bcast(X[1], j), bcast(X[2], j)};
        const double uj = bcast(u, j), vj = bcast(v, j);
        if (ok) {
            double z;
            const double e2 = reproj_e2(Rh, th, k, Xj, uj, vj, &z);
            if (e2 <= 64.0 && z > 1e-9) { ++cnt; se += e2; }
        }
    }
    const Best best = wave_best(Best{ok ? cnt : -1, ok ? se : INFINITY, lane});
    int best_cnt = best.cnt;
    if (best.cnt >= 0) {
#pragma unroll
        for (int i = 0; i < 9; ++i) R[i] = bcast(Rh[i], best.h);
#pragma unroll
        for (int i = 0; i < 3; ++i) t[i] = bcast(th[i], best.h);
    }
    {   // hypothesis NH_PNP: least-squares homography over every z=0 point (stable when a 4-point sample is not)
        double Hl[9], Rl[9], tl[3];
        if (homography_lsq(gmask, X[0], X[1], u, v, 10, Hl) && pose_from_homography(Hl, k.fx, k.fy, k.cx, k.cy, Rl, tl)) {
            refit_pose(sched, gmask, Rl, tl, k, X, u, v);
            double z;
            const double e2 = reproj_e2(Rl, tl, k, X, u, v, &z);
            const bool inl = ((mask >> lane) & 1) && e2 <= 64.0 && z > 1e-9;
            const int c2 = popc64(__ballot(inl));
            const double s2 = wsum(inl ? e2 : 0.0);
            if (c2 > best_cnt || (c2 == best_cnt && s2 < best.s)) {
                best_cnt = c2;
#pragma unroll
                for (int i = 0; i < 9; ++i) R[i] = Rl[i];
                t[0] = tl[0]; t[1] = tl[1]; t[2] = tl[2];
            }
        }
    }
    if (best_cnt < 4) return false;
    double z;
    const double e2 = reproj_e2(R, t, k, X, u, v, &z);
    const u64 inl = __ballot(((mask >> lane) & 1) && e2 <= 64.0 && z > 1e-9);
    refit_pose(sched, inl, R, t, k, X, u, v);
    return true;
}

// ---- calibrateCamera restatement (planar views, pp fixed at ((w-1)/2,(h-1)/2), aspect 1, no distortion) ----
struct View { u64 mask; int kind; double weight; };   // kind 0 ground (x,y), 1 goal plane (y,z)

__device__ bool calibrate_planes(int sched, const View* views, int nviews, const double* X32, double u32, double v32, int img_w,
                                 int img_h, double& f_out, double* R0, double* t0) {
    const int lane = threadIdx.x & 63;
    const double cx = (img_w - 1) * 0.5, cy = (img_h - 1) * 0.5;
    double Hs[3][9];
    double n00 = 0, n01 = 0, n11 = 0, r0 = 0, r1 = 0;
    for (int vi = 0; vi < nviews; ++vi) {
        const double px = views[vi].kind ? X32[1] : X32[0], py = views[vi].kind ? X32[2] : X32[1];
        if (!homography_lsq(views[vi].mask, px, py, u32, v32, 10, Hs[vi])) return false;
        double Hc[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) Hc[i] = Hs[vi][i];
        Hc[0] -= Hc[6] * cx; Hc[1] -= Hc[7] * cx; Hc[2] -= Hc[8] * cx;
        Hc[3] -= Hc[6] * cy; Hc[4] -= Hc[7] * cy; Hc[5] -= Hc[8] * cy;
        double h[3] = {Hc[0], Hc[3], Hc[6]}, v[3] = {Hc[1], Hc[4], Hc[7]}, d1[3], d2[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) { d1[i] = (h[i] + v[i]) * 0.5; d2[i] = (h[i] - v[i]) * 0.5; }
        auto nrm = [](double* a) {
            const double n = fmax(sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]), 1e-300);
            a[0] /= n; a[1] /= n; a[2] /= n;
        };
        nrm(h); nrm(v); nrm(d1); nrm(d2);
        const double sw = sqrt(views[vi].weight);
        const double a0[2] = {sw * (h[0] * v[0]), sw * (h[1] * v[1])}, b0 = -sw * h[2] * v[2];
        const double a1[2] = {sw * (d1[0] * d2[0]), sw * (d1[1] * d2[1])}, b1 = -sw * d1[2] * d2[2];
        n00 += a0[0] * a0[0] + a1[0] * a1[0]; n01 += a0[0] * a0[1] + a1[0] * a1[1]; n11 += a0[1] * a0[1] + a1[1] * a1[1];
        r0 += a0[0] * b0 + a1[0] * b1; r1 += a0[1] * b0 + a1[1] * b1;
    }
    const double det = n00 * n11 - n01 * n01;
    if (!(fabs(det) > 1e-14 * fmax(n00 * n11, 1e-300))) return false;
    const double s0 = (n11 * r0 - n01 * r1) / det, s1 = (n00 * r1 - n01 * r0) / det;
    if (s0 == 0 || s1 == 0) return false;
    double f = 0.5 * (sqrt(fabs(1.0 / s0)) + sqrt(fabs(1.0 / s1)));
    if (!isfinite(f) || f <= 0) return false;
    double Rv[3][9], tv[3][3];
    for (int vi = 0; vi < nviews; ++vi) {
        if (!pose_from_homography(Hs[vi], f, f, cx, cy, Rv[vi], tv[vi])) return false;
        const double Xp[3] = {views[vi].kind ? X32[1] : X32[0], views[vi].kind ? X32[2] : X32[1], 0.0};
        const K4 k{f, f, cx, cy};
        refit_pose(sched, views[vi].mask, Rv[vi], tv[vi], k, Xp, u32, v32);
    }
    if (sched == SCHED_OPENCV) {
        // calibrateCamera's joint fit (cvCalibrateCamera2Internal, CvLevMarq::updateAlt, criteria (30, DBL_EPSILON)): free parameters
        // fy (fx slaved) and [rvec, tvec] per view; duplicated views (Q1) are weights; block-arrowhead normal equations through the
        // Schur complement on f (= OpenCV's dense SVD solve whenever the pose blocks are non-singular)
        double xs[3][6];
        for (int vi = 0; vi < nviews; ++vi) {
            log_so3(Rv[vi], xs[vi]);
            xs[vi][3] = tv[vi][0]; xs[vi][4] = tv[vi][1]; xs[vi][5] = tv[vi][2];
        }
        double A[3][6][6], Bv[3][6], g[3][6], aff = 0, gf = 0;
        auto evaluate = [&](double f_, const double (*xx)[6], bool want_j) -> double {
            if (!(f_ > 0)) return INFINITY;
            double err = 0;
            if (want_j) { aff = 0; gf = 0; }
            for (int vi = 0; vi < nviews; ++vi) {
                const bool in = (views[vi].mask >> lane) & 1;
                const double Xp[3] = {views[vi].kind ? X32[1] : X32[0], views[vi].kind ? X32[2] : X32[1], 0.0};
                double Rr[9], Jl[9], ju[6], jv[6], ru, rv, xn, yn;
                exp_so3(xx[vi], Rr);
                left_jacobian_so3(xx[vi], Jl);
                pose_rows_rvec(Rr, Jl, xx[vi] + 3, f_, f_, cx, cy, Xp, u32, v32, ju, jv, ru, rv, xn, yn);
                const double wgt = views[vi].weight;
                err += wgt * wsum(in ? ru * ru + rv * rv : 0.0);
                if (want_j) {
#pragma unroll
                    for (int i = 0; i < 6; ++i) {
#pragma unroll
                        for (int j = i; j < 6; ++j) {
                            const double s2 = wgt * wsum(in ? ju[i] * ju[j] + jv[i] * jv[j] : 0.0);
                            A[vi][i][j] = s2; A[vi][j][i] = s2;
                        }
                        Bv[vi][i] = wgt * wsum(in ? ju[i] * xn + jv[i] * yn : 0.0);
                        g[vi][i] = wgt * wsum(in ? ju[i] * ru + jv[i] * rv : 0.0);
                    }
                    aff += wgt * wsum(in ? xn * xn + yn * yn : 0.0);
                    gf += wgt * wsum(in ? xn * ru + yn * rv : 0.0);
                }
            }
            return err;
        };
        double e_prev = evaluate(f, xs, true);
        int kk = -3, iters = 0;
        for (;;) {
            double fc = f, xc[3][6], e = INFINITY;
            bool have = false;
            for (;;) {
                const double lam = pow(10.0, (double)kk);
                double s_aff = aff * (1 + lam), s_g = gf, AiB[3][6], Aig[3][6];
                bool ok = true;
                for (int vi = 0; vi < nviews && ok; ++vi) {
                    double Ad[6][6];
#pragma unroll
                    for (int i = 0; i < 6; ++i)
#pragma unroll
                        for (int j = 0; j < 6; ++j) Ad[i][j] = A[vi][i][j] + (i == j ? lam * A[vi][i][i] : 0.0);
                    // (OpenCV: one dense cv::solve(DECOMP_SVD); a pose block that is not positive definite never aborts the step)
                    sym_solve6(Ad, Bv[vi], AiB[vi]);
                    sym_solve6(Ad, g[vi], Aig[vi]);
#pragma unroll
                    for (int i = 0; i < 6; ++i) { s_aff -= Bv[vi][i] * AiB[vi][i]; s_g -= Bv[vi][i] * Aig[vi][i]; }
                }
                have = ok && !(fabs(s_aff) < 1e-300);
                if (have) {
                    const double df = s_g / s_aff;                  // x' = x - d
                    fc = f - df;
                    for (int vi = 0; vi < nviews; ++vi)
#pragma unroll
                        for (int i = 0; i < 6; ++i) xc[vi][i] = xs[vi][i] - (Aig[vi][i] - AiB[vi][i] * df);
                    e = evaluate(fc, xc, false);
                } else e = INFINITY;
                if (!(e > e_prev)) break;
                if (++kk > 16) break;
            }
            if (!have || !isfinite(e)) break;
            kk = max(kk - 1, -16);
            double dn = (fc - f) * (fc - f), pn = f * f;
            for (int vi = 0; vi < nviews; ++vi)
#pragma unroll
                for (int i = 0; i < 6; ++i) { dn += views[vi].weight * (xc[vi][i] - xs[vi][i]) * (xc[vi][i] - xs[vi][i]); pn += views[vi].weight * xs[vi][i] * xs[vi][i]; xs[vi][i] = xc[vi][i]; }
            f = fc;
            ++iters;
            if (iters >= 30 || sqrt(dn) / fmax(sqrt(pn), 1e-300) < DBL_EPS) break;
            e_prev = evaluate(f, xs, true);
        }
        if (!isfinite(f) || f <= 0) return false;
        f_out = f;
        exp_so3(xs[0], R0);
        polar3(R0);
        t0[0] = xs[0][3]; t0[1] = xs[0][4]; t0[2] = xs[0][5];
        return true;
    }
    auto total_cost = [&](double f_, double (*Rs)[9], double (*ts)[3]) {
        double c = 0;
        for (int vi = 0; vi < nviews; ++vi) {
            const double Xp[3] = {views[vi].kind ? X32[1] : X32[0], views[vi].kind ? X32[2] : X32[1], 0.0};
            const K4 k{f_, f_, cx, cy};
            double z;
            const double e2 = reproj_e2(Rs[vi], ts[vi], k, Xp, u32, v32, &z);
            c += views[vi].weight * wsum(((views[vi].mask >> lane) & 1) ? e2 : 0.0);
        }
        return c;
    };
    double lam = 1e-3;
    double c0 = total_cost(f, Rv, tv);
    for (int it = 0; it < 60; ++it) {
        double A[3][6][6], Bv[3][6], g[3][6];
        double aff = 0, gf = 0;
        for (int vi = 0; vi < nviews; ++vi) {
            const bool in = (views[vi].mask >> lane) & 1;
            const double Xp[3] = {views[vi].kind ? X32[1] : X32[0], views[vi].kind ? X32[2] : X32[1], 0.0};
            double ju[6], jv[6], ru, rv, xn, yn;
            pose_rows(Rv[vi], tv[vi], f, f, cx, cy, Xp, u32, v32, ju, jv, ru, rv, xn, yn);
            const double wgt = views[vi].weight;
#pragma unroll
            for (int i = 0; i < 6; ++i) {
#pragma unroll
                for (int j = i; j < 6; ++j) {
                    const double s = wgt * wsum(in ? ju[i] * ju[j] + jv[i] * jv[j] : 0.0);
                    A[vi][i][j] = s; A[vi][j][i] = s;
                }
                Bv[vi][i] = wgt * wsum(in ? ju[i] * xn + jv[i] * yn : 0.0);
                g[vi][i] = wgt * wsum(in ? ju[i] * ru + jv[i] * rv : 0.0);
            }
            aff += wgt * wsum(in ? xn * xn + yn * yn : 0.0);
            gf += wgt * wsum(in ? xn * ru + yn * rv : 0.0);
        }
        bool improved = false;
        double dc = 0;
        for (int tr = 0; tr < 12; ++tr) {
            double s_aff = aff * (1 + lam), s_g = gf;
            double AiB[3][6], Aig[3][6];
            bool ok = true;
            for (int vi = 0; vi < nviews && ok; ++vi) {
                double Ad[6][6];
#pragma unroll
                for (int i = 0; i < 6; ++i)
#pragma unroll
                    for (int j = 0; j < 6; ++j) Ad[i][j] = A[vi][i][j] + (i == j ? lam * A[vi][i][i] : 0.0);
                ok = chol_solve<6>(Ad, Bv[vi], AiB[vi]) && chol_solve<6>(Ad, g[vi], Aig[vi]);
                if (ok) {
#pragma unroll
                    for (int i = 0; i < 6; ++i) { s_aff -= Bv[vi][i] * AiB[vi][i]; s_g -= Bv[vi][i] * Aig[vi][i]; }
                }
            }
            if (!ok || fabs(s_aff) < 1e-300) { lam *= 10; continue; }
            const double df = -s_g / s_aff;
            double Rn[3][9], tn[3][3];
            for (int vi = 0; vi < nviews; ++vi) {
                double step[6];
#pragma unroll
                for (int i = 0; i < 6; ++i) step[i] = -(Aig[vi][i] + AiB[vi][i] * df);
                apply_step(Rv[vi], tv[vi], step, Rn[vi], tn[vi]);
            }
            const double fn = f + df;
            const double c1 = fn > 0 ? total_cost(fn, Rn, tn) : INFINITY;
            if (c1 < c0) {
                dc = c0 - c1; c0 = c1; f = fn;
                for (int vi = 0; vi < nviews; ++vi) {
#pragma unroll
                    for (int i = 0; i < 9; ++i) Rv[vi][i] = Rn[vi][i];
                    tv[vi][0] = tn[vi][0]; tv[vi][1] = tn[vi][1]; tv[vi][2] = tn[vi][2];
                }
                lam = fmax(lam * 0.1, 1e-15);
                improved = true;
                break;
            }
            lam *= 10;
        }
        if (!improved || dc <= 1e-16 * fmax(c0, 1e-30)) break;
    }
    f_out = f;
#pragma unroll
    for (int i = 0; i < 9; ++i) R0[i] = Rv[0][i];
    polar3(R0);
    t0[0] = tv[0][0]; t0[1] = tv[0][1]; t0[2] = tv[0][2];
    return true;
}

// ---- camera record + reference control flow ----------------------------------------------------------
struct Cam {
    double R[9], pos[3];
    double fx, fy, cx, cy;      // calibration matrix (cx,cy as left by calibrateCamera: quirk Q3)
    double ppx, ppy;            // principal_point used by project_point / JSON
    double rmse;
    int tag;
};

__device__ __forceinline__ void cam_set_pose(Cam& c, const double* R, const double* t) {   // position = -R^T t
#pragma unroll
    for (int i = 0; i < 9; ++i) c.R[i] = R[i];
    c.pos[0] = -(R[0] * t[0] + R[3] * t[1] + R[6] * t[2]);
    c.pos[1] = -(R[1] * t[0] + R[4] * t[1] + R[7] * t[2]);
    c.pos[2] = -(R[2] * t[0] + R[5] * t[1] + R[8] * t[2]);
}
__device__ __forceinline__ void cam_t(const Cam& c, double* t) {   // t = -R pos
    t[0] = -(c.R[0] * c.pos[0] + c.R[1] * c.pos[1] + c.R[2] * c.pos[2]);
    t[1] = -(c.R[3] * c.pos[0] + c.R[4] * c.pos[1] + c.R[5] * c.pos[2]);
    t[2] = -(c.R[6] * c.pos[0] + c.R[7] * c.pos[1] + c.R[8] * c.pos[2]);
}

struct Pts {   // lane-local point data
    double X64[3], X32[3];
    double u, v, u32, v32;
    int sched;       // SCHED_OPENCV / SCHED_CONVERGED
    int refine_iters;   // cap of refine_camera's LMSolver run (sncal_voter_cfg.refine_max_iters)
};

__device__ bool cam_solve_pnp(Cam& c, u64 mask, const Pts& p) {
    double R[9], t[3];
    const K4 k{c.fx, c.fy, c.cx, c.cy};
    if (!pnp_ransac(p.sched, mask, mask & GROUND_MASK, k, p.X64, p.u, p.v, R, t)) return false;
    cam_set_pose(c, R, t);
    return true;
}
__device__ void cam_refine(Cam& c, u64 mask, const Pts& p) {
    double R[9], t[3];
#pragma unroll
    for (int i = 0; i < 9; ++i) R[i] = c.R[i];
    cam_t(c, t);
    const K4 k{c.fx, c.fy, c.cx, c.cy};
    if (p.sched == SCHED_OPENCV) lm_solver_pose_auto(mask, R, t, k, p.X64, p.u, p.v, p.refine_iters, 1e-5);      // camera.py:116-117
    else refine_pose_lm(mask, R, t, k, p.X64, p.u, p.v, 100, 1e-10);
    cam_set_pose(c, R, t);
}
// Camera.projection_rmse (camera.py:270-277; project_point :249-268 with the fp32 round trip of distort :247)
__device__ double cam_rmse(const Cam& c, u64 mask, const Pts& p) {
    const int lane = threadIdx.x & 63;
    const double d[3] = {p.X64[0] - c.pos[0], p.X64[1] - c.pos[1], p.X64[2] - c.pos[2]};
    double r[3];
    mul3v(c.R, d, r);
    double px = 0, py = 0;
    if (!(r[2] <= 1e-3)) {
        const float xn = (float)(r[0] / r[2]), yn = (float)(r[1] / r[2]);
        px = (double)xn * c.fx + c.ppx;
        py = (double)yn * c.fy + c.ppy;
    }
    const double l2 = sqrt((p.u - px) * (p.u - px) + (p.v - py) * (p.v - py));
    return wsum(((mask >> lane) & 1) ? l2 : 0.0) / (double)popc64(mask);
}
__device__ __forceinline__ bool good_camera(const Cam& c) {   // prediction.py:469-484
    return c.fx >= 10 && c.fx <= 20000 && c.pos[0] > -250 && c.pos[0] < 250 && c.pos[1] > -250 && c.pos[1] < 250 &&
           c.pos[2] > -100 && c.pos[2] < 0;
}

__device__ int build_views(u64 mask, int min_pts, bool duplicate, View* views) {
    int nv = 0;
    const u64 pm[3] = {mask & GROUND_MASK, mask & GOAL_LEFT_MASK, mask & GOAL_RIGHT_MASK};
    for (int pl = 0; pl < 3; ++pl) {
        if (!pm[pl]) continue;
        double mult = 1.0;
        if (duplicate) {   // quirk Q1: the list object is appended once per id from the first detected one on
            int first = 0, len = 0;
            if (pl == 0) {
                len = 54;                                            // range(58) minus the 4 crossbar ids (Q6)
                const int id = __ffsll((long long)pm[0]) - 1;
                first = id - popc64(TOP_GATES_MASK & ((1ull << id) - 1));
            } else {
                len = 10;
                const int* ids = pl == 1 ? c_goal_left_ids : c_goal_right_ids;
                first = 10;
                for (int q = 9; q >= 0; --q) if ((pm[pl] >> ids[q]) & 1) first = q;
            }
            mult = (double)(len - first);
        }
        if (popc64(pm[pl]) >= min_pts) { views[nv].mask = pm[pl]; views[nv].kind = pl == 0 ? 0 : 1; views[nv].weight = mult; ++nv; }
    }
    return nv;
}

__device__ void cam_from_calibration(Cam& c, double f, const double* R0, const double* t0, int img_w, int img_h) {
    c.fx = c.fy = f;
    c.cx = (img_w - 1) * 0.5; c.cy = (img_h - 1) * 0.5;
    c.ppx = img_w / 2.0; c.ppy = img_h / 2.0;
    cam_set_pose(c, R0, t0);
}

enum { ST_OK = 0, ST_NONE = 1, ST_RAISE = 2 };   // value / None / exception

// prediction.py:487-520
__device__ int camera_from_homography(u64 mask, const Pts& p, int img_w, int img_h, Cam& c) {
    const u64 g = mask & GROUND_MASK;
    if (popc64(g) < 4) return ST_NONE;
    double H[9];
    if (!homography_ransac(g, p.X32[0], p.X32[1], p.u32, p.v32, 10.0, H)) return ST_NONE;
    double fx, fy;
    if (k_from_homography(H, img_w / 2.0, img_h / 2.0, fx, fy)) {
        c.fx = fx; c.fy = fy; c.cx = img_w / 2.0; c.cy = img_h / 2.0; c.ppx = c.cx; c.ppy = c.cy;
    } else {
        // prediction.py:514 ignores the failure flag of estimate_calibration_matrix_from_plane_homography: the Camera() keeps its
        // initial state -- calibration = eye(3), focal lengths 1 (camera.py:33-40), principal point (w/2, h/2) for project_point -- and
        // goes through solve_pnp / refine_camera / projection_rmse like any other.  Followed since round 4 (rounds 1-3 returned None
        // here): the K = I camera itself never survives the rmse tests of its callers, but a solve_pnp failure under it raises, and
        // the reference then has no camera for the frame.
        c.fx = c.fy = 1.0; c.cx = c.cy = 0.0; c.ppx = img_w / 2.0; c.ppy = img_h / 2.0;
    }
#ifdef SNCAL_SOLVE_TIMING
    const unsigned long long th0 = __builtin_amdgcn_s_memtime();
#endif
    if (!cam_solve_pnp(c, mask, p)) return ST_RAISE;
#ifdef SNCAL_SOLVE_TIMING
    const unsigned long long th1 = __builtin_amdgcn_s_memtime();
#endif
    cam_refine(c, mask, p);
#ifdef SNCAL_SOLVE_TIMING
    if ((threadIdx.x & 63) == 0) printf("  hom wave %d: solve_pnp %llu clk refine %llu clk fx %g\n", (int)(threadIdx.x >> 6), th1 - th0, __builtin_amdgcn_s_memtime() - th1, c.fx);
#endif
    c.rmse = cam_rmse(c, mask, p);
    return ST_OK;
}

// prediction.py:523-555 + get_camera_gen :609-640 (exceptions inside are swallowed -> None)
__device__ int camera_all_points(u64 mask, const Pts& p, int img_w, int img_h, Cam& c) {
    View views[3];
    const int nv = build_views(mask, 6, true, views);
    double total = 0;
    for (int i = 0; i < nv; ++i) total += views[i].weight * popc64(views[i].mask);
    if (!(nv > 0 && total > 6)) return ST_NONE;
    double f, R0[9], t0[3];
#ifdef SNCAL_SOLVE_TIMING
    const unsigned long long tq0 = __builtin_amdgcn_s_memtime();
    const bool cal_ok = calibrate_planes(p.sched, views, nv, p.X32, p.u32, p.v32, img_w, img_h, f, R0, t0);
    const unsigned long long tq1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) printf("  cap wave %d npts %d nviews %d: calibrate_planes %llu clk ok %d f %g\n", (int)(threadIdx.x >> 6), popc64(mask), nv, tq1 - tq0, (int)cal_ok, cal_ok ? f : 0.0);
    if (!cal_ok) return ST_NONE;
    cam_from_calibration(c, f, R0, t0, img_w, img_h);
    const bool pnp_ok = cam_solve_pnp(c, mask, p);
    const unsigned long long tq2 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) printf("  cap wave %d: solve_pnp %llu clk ok %d\n", (int)(threadIdx.x >> 6), tq2 - tq1, (int)pnp_ok);
    if (!pnp_ok) return ST_NONE;
    if (popc64(mask) > 6 && c.fx >= 10 && c.fx <= 20000) cam_refine(c, mask, p);
    const unsigned long long tq3 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) printf("  cap wave %d: refine %llu clk\n", (int)(threadIdx.x >> 6), tq3 - tq2);
    c.rmse = cam_rmse(c, mask, p);
    return ST_OK;
#endif
    if (!calibrate_planes(p.sched, views, nv, p.X32, p.u32, p.v32, img_w, img_h, f, R0, t0)) return ST_NONE;
    cam_from_calibration(c, f, R0, t0, img_w, img_h);
    if (!cam_solve_pnp(c, mask, p)) return ST_NONE;            // always runs (quirk Q2)
    // Same outcome, less work (shared with oracle/solve.py): every caller keeps this camera only if good_camera accepts it, and the
    // focal-length clause does not depend on the pose -- a candidate calibrated outside [10, 20000] px is discarded whatever
    // refine_camera does to it, so it is not refined (under f ~ 0.04 px the reference's 20000-iteration LM runs to the end: 100 ms of
    // one wavefront for a camera nobody uses, which is what the 200-iteration cap of rounds 1-3 was for)
    if (popc64(mask) > 6 && c.fx >= 10 && c.fx <= 20000) cam_refine(c, mask, p);
    c.rmse = cam_rmse(c, mask, p);
    return ST_OK;
}

// prediction.py:572-606
__device__ int camera_accurate_points(u64 mask, const Pts& p, double thr, int img_w, int img_h, Cam& c) {
    const int lane = threadIdx.x & 63;
    const u64 g = mask & GROUND_MASK;
    if (popc64(g) < 4) return ST_NONE;
    double H[9];
    if (!homography_ransac(g, p.X32[0], p.X32[1], p.u32, p.v32, thr, H)) return ST_NONE;
    double pu, pv;
    apply_h(H, p.X32[0], p.X32[1], pu, pv);
    const double err = sqrt((pu - p.u32) * (pu - p.u32) + (pv - p.v32) * (pv - p.v32));
    const u64 sel = __ballot(((g >> lane) & 1) && err < thr) | (mask & TOP_GATES_MASK);
    return camera_all_points(sel, p, img_w, img_h, c);
}

__device__ u64 add_line_points(u64 mask, Pts& p, const float* line_pts, const sncal_voter_cfg& cfg, int mode,
                               int n_ground_kp) {
    // prediction.py:186-192 (mode 2), :270-278 (mode 1, voter), :356-364 (mode 0, original_voter)
    if (!line_pts) return mask;
    const int lane = threadIdx.x & 63;
    for (int i = 0; i < 30; ++i) {
        const float lx = line_pts[i * 3 + 0], ly = line_pts[i * 3 + 1], valid = line_pts[i * 3 + 2];
        if (!(valid > 0.5f) || ((mask >> i) & 1)) continue;
        bool take;
        if (mode == 0) take = n_ground_kp < cfg.min_points_per_plane || (0 <= lx && lx <= cfg.img_w && 0 <= ly && ly <= cfg.img_h);
        else if (mode == 1) take = popc64(mask & GROUND_MASK) < cfg.min_points_per_plane;
        else take = popc64(mask) <= cfg.min_points;
        if (take) {
            mask |= 1ull << i;
            if (lane == i) { p.u = (double)lx; p.v = (double)ly; p.u32 = (double)lx; p.v32 = (double)ly; }
        }
    }
    return mask;
}

__device__ u64 select_points(const float conf, double thr, bool reliable_rule, int reliable_thresh) {
    const int lane = threadIdx.x & 63;
    const bool det = lane < NPTS && (double)conf > thr;
    const u64 dm = __ballot(det);
    if (!reliable_rule || popc64(dm) < reliable_thresh) return dm;
    return dm & KEEP_MASK;
}

// prediction.py:339-437, in the three pieces calibrate_kernel runs on two wavefronts (the homography camera and the calibrated camera are
// independent solves of the same points; the reference builds them one after the other):
//   ov_points   the selection (:345-357)                                  -> mask, and the line points in p
//   ov_hom      camera_from_homography (:359)                             -> hs, hom
//   ov_cal      the multi-plane calibration branch (:361-420)             -> ST_RAISE / ST_OK (a camera, refined) / ST_NONE
//   ov_combine  the reference's order of precedence (:359-437): an exception of either half leaves, the calibrated camera wins, the
//               homography camera is the fallback below rmse 26
__device__ u64 ov_points(const float* kp, const float* line_pts, const sncal_voter_cfg& cfg, double thr, Pts& p) {
    const u64 mask = select_points(kp[2], thr, true, cfg.reliable_thresh);
    return add_line_points(mask, p, line_pts, cfg, 0, popc64(mask & GROUND_MASK));
}
__device__ int ov_cal(u64 mask, const sncal_voter_cfg& cfg, const Pts& p, Cam& out) {
    View views[3];
    const int nv = build_views(mask, cfg.min_points_per_plane, false, views);
    if (!(nv > 0 && popc64(mask) > cfg.min_points)) return ST_NONE;
    double f, R0[9], t0[3];
    if (!calibrate_planes(p.sched, views, nv, p.X32, p.u32, p.v32, cfg.img_w, cfg.img_h, f, R0, t0)) return ST_RAISE;
    cam_from_calibration(out, f, R0, t0, cfg.img_w, cfg.img_h);
    out.tag = SNCAL_CAM_ORIGINAL;
    if (popc64(mask & GROUND_MASK) < cfg.min_points_per_plane && !cam_solve_pnp(out, mask, p)) return ST_RAISE;
    if (!good_camera(out)) return ST_NONE;
    if (popc64(mask) > cfg.min_points_for_refinement) cam_refine(out, mask, p);
    return ST_OK;
}
__device__ int ov_combine(u64 mask, const Pts& p, int hs, const Cam& hom, int cs, const Cam& cal, Cam& out) {
    if (hs == ST_RAISE || cs == ST_RAISE) return ST_RAISE;      // (the serial order raises in the homography half first: same outcome)
    if (cs == ST_OK) out = cal;
    else if (hs == ST_OK && hom.rmse < 26) { out = hom; out.tag = SNCAL_CAM_ORIGINAL_HOM; }
    else return ST_NONE;
    out.rmse = cam_rmse(out, mask, p);
    return ST_OK;
}
__device__ int original_voter(const float* kp, const float* line_pts, const sncal_voter_cfg& cfg, double thr, Pts p, Cam& out) {
    const u64 mask = ov_points(kp, line_pts, cfg, thr, p);
    Cam hom, cal;
    const int hs = camera_from_homography(mask, p, cfg.img_w, cfg.img_h, hom);
    if (hs == ST_RAISE) return ST_RAISE;
    const int cs = ov_cal(mask, cfg, p, cal);
    return ov_combine(mask, p, hs, hom, cs, cal, out);
}

// prediction.py:259-330
// the final choice among the homography camera and the four subset cameras (prediction.py:293-329)
__device__ int voter_select(const sncal_voter_cfg& cfg, int hs, const Cam& hom, const int (&st)[4], const Cam (&cands)[4], Cam& out) {
    if (hs == ST_RAISE) return ST_RAISE;
    const int tags[4] = {SNCAL_CAM_VOTER_REL, SNCAL_CAM_VOTER_ACC, SNCAL_CAM_VOTER_ALL, SNCAL_CAM_VOTER_GROUND};
    int best = -1;
    bool best_flag = false;
    double best_inv = 0;
    for (int i = 0; i < 4; ++i) {          // python max(): first maximum of (flag, 1/rmse) in list order
        if (st[i] != ST_OK || !good_camera(cands[i])) continue;
        if (cands[i].rmse == 0.0) return ST_RAISE;                    // 1/0 -> ZeroDivisionError (quirk Q5)
        const bool flag = i == 0 && cands[i].rmse < cfg.max_rmse_rel;
        const double inv = 1.0 / cands[i].rmse;
        if (best < 0 || (flag && !best_flag) || (flag == best_flag && inv > best_inv)) { best = i; best_flag = flag; best_inv = inv; }
    }
    if (best >= 0 && cands[best].rmse < cfg.max_rmse) { out = cands[best]; out.tag = tags[best]; return ST_OK; }
    if (hs == ST_OK && hom.rmse < cfg.max_rmse) { out = hom; out.tag = SNCAL_CAM_VOTER_HOM; return ST_OK; }
    return ST_NONE;
}

__device__ int voter(const float* kp, const float* line_pts, const sncal_voter_cfg& cfg, double thr, Pts p, Cam& out) {
    u64 mask = select_points(kp[2], thr, false, 0);
    mask = add_line_points(mask, p, line_pts, cfg, 1, 0);
    Cam hom;
    const int hs = camera_from_homography(mask, p, cfg.img_w, cfg.img_h, hom);
    if (hs == ST_RAISE) return ST_RAISE;
    Cam cands[4];
    int st[4];
    st[2] = camera_all_points(mask, p, cfg.img_w, cfg.img_h, cands[2]);
    st[0] = camera_all_points(mask & KEEP_MASK, p, cfg.img_w, cfg.img_h, cands[0]);
    st[1] = camera_accurate_points(mask, p, 5.0, cfg.img_w, cfg.img_h, cands[1]);
    st[3] = camera_all_points(mask & GROUND_MASK, p, cfg.img_w, cfg.img_h, cands[3]);
    return voter_select(cfg, hs, hom, st, cands, out);
}

// The same voter spread over the four waves of a workgroup: its five cameras are independent solves of the same
// points (prediction.py:281-291 builds them one after the other), so wave 0 takes the homography camera and the
// ground-plane subset, waves 1..3 the all / reliable / H-consistent subsets; every wave then runs the (cheap) selection
// on the five results in LDS.  Each camera is computed by the same code on the same inputs as in voter(): identical bits.
struct VoterShared { Cam hom; Cam cands[4]; int hs; int st[4]; };

__device__ int voter_parallel(const float* kp, const float* line_pts, const sncal_voter_cfg& cfg, double thr, Pts p, VoterShared& sh,
                              Cam& out) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    u64 mask = select_points(kp[2], thr, false, 0);
    mask = add_line_points(mask, p, line_pts, cfg, 1, 0);
    Cam c;
    c.tag = SNCAL_CAM_NONE;
    if (wave == 0) {
        const int hs = camera_from_homography(mask, p, cfg.img_w, cfg.img_h, c);
        if (lane == 0) { sh.hom = c; sh.hs = hs; }
        const int s3 = camera_all_points(mask & GROUND_MASK, p, cfg.img_w, cfg.img_h, c);
        if (lane == 0) { sh.cands[3] = c; sh.st[3] = s3; }
    } else if (wave == 1) {
        const int s2 = camera_all_points(mask, p, cfg.img_w, cfg.img_h, c);
        if (lane == 0) { sh.cands[2] = c; sh.st[2] = s2; }
    } else if (wave == 2) {
        const int s0 = camera_all_points(mask & KEEP_MASK, p, cfg.img_w, cfg.img_h, c);
        if (lane == 0) { sh.cands[0] = c; sh.st[0] = s0; }
    } else {
        const int s1 = camera_accurate_points(mask, p, 5.0, cfg.img_w, cfg.img_h, c);
        if (lane == 0) { sh.cands[1] = c; sh.st[1] = s1; }
    }
    __syncthreads();
    const int r = voter_select(cfg, sh.hs, sh.hom, sh.st, sh.cands, out);
    __syncthreads();                                   // everyone has read the results before the next pass overwrites them
    return r;
}

// prediction.py:138-170
__device__ int opencv_calibration(const float* kp, const sncal_voter_cfg& cfg, const Pts& p, Cam& out) {
    const int lane = threadIdx.x & 63;
    const u64 mask = __ballot(lane < NPTS && (double)kp[2] > cfg.conf_thresh) & GROUND_MASK;
    if (popc64(mask) <= 5) return ST_NONE;
    View v{mask, 0, 1.0};
    double f, R0[9], t0[3];
    if (!calibrate_planes(p.sched, &v, 1, p.X32, p.u32, p.v32, cfg.img_w, cfg.img_h, f, R0, t0)) return ST_RAISE;
    cam_from_calibration(out, f, R0, t0, cfg.img_w, cfg.img_h);
    out.tag = SNCAL_CAM_ORIGINAL;
    out.rmse = cam_rmse(out, mask, p);
    return ST_OK;
}

// prediction.py:172-243
__device__ int opencv_calibration_multiplane(const float* kp, const float* line_pts, const sncal_voter_cfg& cfg, Pts p, Cam& out) {
    u64 mask = select_points(kp[2], cfg.conf_thresh, true, cfg.reliable_thresh);
    mask = add_line_points(mask, p, line_pts, cfg, 2, 0);
    View views[3];
    const int nv = build_views(mask, cfg.min_points_per_plane, false, views);
    if (!(nv > 0 && popc64(mask) > cfg.min_points)) return ST_NONE;
    double f, R0[9], t0[3];
    if (!calibrate_planes(p.sched, views, nv, p.X32, p.u32, p.v32, cfg.img_w, cfg.img_h, f, R0, t0)) return ST_RAISE;
    if (!(f > cfg.min_focal_length)) return ST_NONE;
    cam_from_calibration(out, f, R0, t0, cfg.img_w, cfg.img_h);
    if (popc64(mask) > cfg.min_points_for_refinement) cam_refine(out, mask, p);
    out.tag = SNCAL_CAM_ORIGINAL;
    out.rmse = cam_rmse(out, mask, p);
    return ST_OK;
}

__device__ void load_points(const float* kp, Pts& p) {
    const int lane = threadIdx.x & 63;
    const int id = lane < NPTS ? lane : 0;
#pragma unroll
    for (int i = 0; i < 3; ++i) { p.X64[i] = c_P64[id * 3 + i]; p.X32[i] = c_P32[id * 3 + i]; }
    p.u = (double)kp[0]; p.v = (double)kp[1];      // float(pred[i,0]) -> python float; float32 -> float64 is exact
    p.u32 = p.u; p.v32 = p.v;
}

// Four wavefronts per workgroup (calibrate_kernel: four frames, or two frames x two cameras): a solver wave owns a whole SIMD register
// file (512 VGPRs), so a lone wave per CU would keep the co-running convolution workgroups (one wave on each SIMD) off that CU; packed,
// 64 frames block 16-32 CUs instead of degrading 64.
constexpr int STATUS_PENDING = -1;      // iterative_voter frames whose original_voter pass found no camera: left for voter_kernel

__device__ __forceinline__ void store_camera(sncal_camera* out, int st, const Cam& cam) {
    sncal_camera o;
    memset(&o, 0, sizeof(o));
    if (st == ST_OK) {
        for (int i = 0; i < 3; ++i) o.position[i] = cam.pos[i];
        for (int i = 0; i < 9; ++i) o.rotation[i] = cam.R[i];
        o.fx = cam.fx; o.fy = cam.fy; o.cx = cam.cx; o.cy = cam.cy; o.rmse = cam.rmse;
        o.status = cam.tag;
    }
    *out = o;
}

// iterative_voter's second half (prediction.py:250-256) for the frames calibrate_kernel left pending: one workgroup per
// frame, the voter's five cameras on four waves.  Frames that already have a camera leave at once.
__global__ __launch_bounds__(256, 1) void voter_kernel(const float* __restrict__ kpts, const float* __restrict__ line_pts, int B,
                                                       sncal_voter_cfg cfg, sncal_camera* __restrict__ out) {
    __shared__ VoterShared sh;
    const int frame = blockIdx.x;
    if (out[frame].status != STATUS_PENDING) return;
    const int lane = threadIdx.x & 63;
    float kp[3] = {0.f, 0.f, -1.f};
    if (lane < NPTS) {
        const float* src = kpts + ((size_t)frame * NPTS + lane) * 3;
        kp[0] = src[0]; kp[1] = src[1]; kp[2] = src[2];
    }
    const float* lp = line_pts ? line_pts + (size_t)frame * 90 : nullptr;
    Pts p;
    load_points(kp, p);
    p.sched = cfg.lm_schedule == 1 ? SCHED_CONVERGED : SCHED_OPENCV;
    p.refine_iters = cfg.refine_max_iters > 0 ? cfg.refine_max_iters : 20000;
    Cam cam;
    cam.tag = SNCAL_CAM_NONE;
    int st = ST_NONE;
    for (int i = 0; i < cfg.n_conf_threshs; ++i) {
        st = voter_parallel(kp, lp, cfg, cfg.conf_threshs[i], p, sh, cam);
        if (st != ST_NONE) break;                      // camera found, or an exception leaves iterative_voter
    }
    if (threadIdx.x == 0) store_camera(out + frame, st, cam);
}

// The same second half with ONE WAVEFRONT PER CAMERA (round 4).  A frame that ends without a camera walks all of iterative_voter's
// thresholds, and every threshold costs the voter's five cameras: on voter_kernel's four waves (wave 0 owns two cameras) the bench's
// slowest frame took 2 x 5.9 ms = the whole 11.7 ms of the launch -- eight weak points, f = 53 px, every camera 1.1 ms of
// calibrate_planes + 0.6 ms of RANSAC PnP + 2 ms of refine_camera at the bench's 200-iteration cap.  The cameras of ALL thresholds are
// independent of each other (prediction.py:250-256 only stops at the first threshold that yields one; what a threshold computes does
// not depend on the thresholds before it), so they are computed side by side, threshold x camera = up to 15 wavefronts per pending
// frame, each on the code and the inputs the serial voter gives it (identical bits), and voter_select_kernel then walks the thresholds
// in the reference's order and keeps the first result that is not "no camera".  Wall time = the slowest single camera (3.8 ms on
// that frame); the work of thresholds behind the deciding one is wasted CU time on a few wavefronts.
enum { VT_HOM = 0, VT_GROUND = 1, VT_ALL = 2, VT_KEEP = 3, VT_ACC = 4, VT_TASKS = 5 };
__global__ __launch_bounds__(64, 1) void voter_task_kernel(const float* __restrict__ kpts, const float* __restrict__ line_pts, int B,
                                                           sncal_voter_cfg cfg, const sncal_camera* __restrict__ out, VoterShared* __restrict__ slots) {
    const int task = (int)blockIdx.x % VT_TASKS, ti = ((int)blockIdx.x / VT_TASKS) % cfg.n_conf_threshs;
    const int frame = (int)blockIdx.x / (VT_TASKS * cfg.n_conf_threshs);
    if (frame >= B || out[frame].status != STATUS_PENDING) return;
    const int lane = threadIdx.x & 63;
    float kp[3] = {0.f, 0.f, -1.f};
    if (lane < NPTS) {
        const float* src = kpts + ((size_t)frame * NPTS + lane) * 3;
        kp[0] = src[0]; kp[1] = src[1]; kp[2] = src[2];
    }
    const float* lp = line_pts ? line_pts + (size_t)frame * 90 : nullptr;
    Pts p;
    load_points(kp, p);
    p.sched = cfg.lm_schedule == 1 ? SCHED_CONVERGED : SCHED_OPENCV;
    p.refine_iters = cfg.refine_max_iters > 0 ? cfg.refine_max_iters : 20000;
    u64 mask = select_points(kp[2], cfg.conf_threshs[ti], false, 0);
    mask = add_line_points(mask, p, lp, cfg, 1, 0);
    VoterShared& sh = slots[(size_t)frame * cfg.n_conf_threshs + ti];
    Cam c;
    c.tag = SNCAL_CAM_NONE;
    if (task == VT_HOM) {
        const int hs = camera_from_homography(mask, p, cfg.img_w, cfg.img_h, c);
        if (lane == 0) { sh.hom = c; sh.hs = hs; }
    } else {
        int s = ST_NONE, slot = 0;
        if (task == VT_GROUND) { s = camera_all_points(mask & GROUND_MASK, p, cfg.img_w, cfg.img_h, c); slot = 3; }
        else if (task == VT_ALL) { s = camera_all_points(mask, p, cfg.img_w, cfg.img_h, c); slot = 2; }
        else if (task == VT_KEEP) { s = camera_all_points(mask & KEEP_MASK, p, cfg.img_w, cfg.img_h, c); slot = 0; }
        else { s = camera_accurate_points(mask, p, 5.0, cfg.img_w, cfg.img_h, c); slot = 1; }
        if (lane == 0) { sh.cands[slot] = c; sh.st[slot] = s; }
    }
}

// prediction.py:250-256 on the cameras voter_task_kernel left: thresholds in order, the first one whose voter returns a camera (or raises)
// ends the frame.  One thread per frame (the selection is scalar code).
__global__ __launch_bounds__(64) void voter_select_kernel(int B, sncal_voter_cfg cfg, const VoterShared* __restrict__ slots, sncal_camera* __restrict__ out) {
    const int frame = (int)(blockIdx.x * 64 + threadIdx.x);
    if (frame >= B || out[frame].status != STATUS_PENDING) return;
    Cam cam;
    cam.tag = SNCAL_CAM_NONE;
    int st = ST_NONE;
    for (int i = 0; i < cfg.n_conf_threshs; ++i) {
        const VoterShared& sh = slots[(size_t)frame * cfg.n_conf_threshs + i];
        st = voter_select(cfg, sh.hs, sh.hom, sh.st, sh.cands, cam);
        if (st != ST_NONE) break;
    }
    store_camera(out + frame, st, cam);
}

// iterative_voter / original_voter (algorithms 0, 1): TWO wavefronts per frame, two frames per workgroup -- wave 2 i the homography
// camera, wave 2 i + 1 the calibrated camera of frame i (ov_hom / ov_cal above), the even wave combines them in the reference's order
// and goes on (pending mark, or the serial voter when the second stage is switched off).  The other algorithms: one wavefront per frame,
// four frames per workgroup.  The launcher sizes the grid accordingly.
struct OvHalf { Cam cam; int st; };
__global__ __launch_bounds__(256, 1) void calibrate_kernel(const float* __restrict__ kpts, const float* __restrict__ line_pts,
                                                           int B, sncal_voter_cfg cfg, sncal_camera* __restrict__ out, int defer_voter) {
    __shared__ OvHalf half[2];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const bool paired = cfg.algorithm <= 1;
    const int frame = paired ? (int)blockIdx.x * 2 + (wave >> 1) : (int)blockIdx.x * (int)(blockDim.x >> 6) + wave;
    const int role = paired ? wave & 1 : 0;
    const bool valid = frame < B;
    if (!paired && !valid) return;
    float kp[3] = {0.f, 0.f, -1.f};
    if (valid && lane < NPTS) {
        const float* src = kpts + ((size_t)frame * NPTS + lane) * 3;
        kp[0] = src[0]; kp[1] = src[1]; kp[2] = src[2];
    }
    const float* lp = line_pts && valid ? line_pts + (size_t)frame * 90 : nullptr;
    Pts p;
    load_points(kp, p);
    p.sched = cfg.lm_schedule == 1 ? SCHED_CONVERGED : SCHED_OPENCV;
    p.refine_iters = cfg.refine_max_iters > 0 ? cfg.refine_max_iters : 20000;
    Cam cam;
    cam.tag = SNCAL_CAM_NONE;
    int st = ST_NONE;
    if (paired) {
        u64 mask = 0;
        Cam hom;
        int hs = ST_NONE;
        if (valid) {
            mask = ov_points(kp, lp, cfg, cfg.algorithm == 0 ? 0.5 : cfg.conf_thresh, p);
            if (role == 1) {
                Cam cal;
                cal.tag = SNCAL_CAM_NONE;
                const int cs = ov_cal(mask, cfg, p, cal);
                if (lane == 0) { half[wave >> 1].cam = cal; half[wave >> 1].st = cs; }
            } else {
                hs = camera_from_homography(mask, p, cfg.img_w, cfg.img_h, hom);
            }
        }
        __syncthreads();
        if (!valid || role == 1) return;
        st = ov_combine(mask, p, hs, hom, half[wave >> 1].st, half[wave >> 1].cam, cam);
        if (cfg.algorithm == 0 && st != ST_OK) {   // iterative_voter, prediction.py:245-257
            if (defer_voter) {
                if (lane == 0) { store_camera(out + frame, ST_NONE, cam); out[frame].status = STATUS_PENDING; }
                return;
            }
            st = ST_NONE;
            for (int i = 0; i < cfg.n_conf_threshs; ++i) {
                st = voter(kp, lp, cfg, cfg.conf_threshs[i], p, cam);
                if (st != ST_NONE) break;   // camera found, or an exception leaves iterative_voter
            }
        }
    } else {
        switch (cfg.algorithm) {
            case 2: st = voter(kp, lp, cfg, cfg.conf_thresh, p, cam); break;
            case 3: st = opencv_calibration(kp, cfg, p, cam); break;
            default: st = opencv_calibration_multiplane(kp, lp, cfg, p, cam); break;
        }
    }
    if (lane == 0) store_camera(out + frame, st, cam);
}

// Round 5: the first pass as SINGLE-WAVE workgroups.  calibrate_kernel's paired form couples the two halves of a frame (and two frames)
// in one 256-thread workgroup through LDS and a barrier: a workgroup of four 512-register waves needs a completely free CU, and it
// keeps all four SIMDs until its slowest wave is through -- at the reference's refine criterion that can be a 330 ms Levenberg-Marquardt
// crawl.  On the CU-masked solve streams of the pipeline (8 CUs, pipeline.py) the crawling waves of earlier batches sit one per CU, no
// CU ever has four free SIMDs, and the next batch's first pass waited for them: the step went from 113 to 155 ms (measured).  Here every
// (frame, half) is a 64-thread workgroup that needs ONE free SIMD and leaves its result in a per-frame slot; first_pass_combine_kernel
// (one wave per frame, small) then applies the reference's order of precedence.  Same device functions on the same inputs as the
// paired form: identical bytes (tests/test_solve_gpu.py).
struct FirstPass { Cam hom; Cam cal; int hs; int cs; };
__global__ __launch_bounds__(64, 1) void first_pass_task_kernel(const float* __restrict__ kpts, const float* __restrict__ line_pts, int B,
                                                                sncal_voter_cfg cfg, FirstPass* __restrict__ fp) {
    const int frame = (int)blockIdx.x >> 1, role = (int)blockIdx.x & 1, lane = threadIdx.x & 63;
    if (frame >= B) return;
    float kp[3] = {0.f, 0.f, -1.f};
    if (lane < NPTS) {
        const float* src = kpts + ((size_t)frame * NPTS + lane) * 3;
        kp[0] = src[0]; kp[1] = src[1]; kp[2] = src[2];
    }
    const float* lp = line_pts ? line_pts + (size_t)frame * 90 : nullptr;
    Pts p;
    load_points(kp, p);
    p.sched = cfg.lm_schedule == 1 ? SCHED_CONVERGED : SCHED_OPENCV;
    p.refine_iters = cfg.refine_max_iters > 0 ? cfg.refine_max_iters : 20000;
    const u64 mask = ov_points(kp, lp, cfg, cfg.algorithm == 0 ? 0.5 : cfg.conf_thresh, p);
    Cam c;
    c.tag = SNCAL_CAM_NONE;
    if (role == 1) {
        const int cs = ov_cal(mask, cfg, p, c);
        if (lane == 0) { fp[frame].cal = c; fp[frame].cs = cs; }
    } else {
        const int hs = camera_from_homography(mask, p, cfg.img_w, cfg.img_h, c);
        if (lane == 0) { fp[frame].hom = c; fp[frame].hs = hs; }
    }
}
__global__ __launch_bounds__(64) void first_pass_combine_kernel(const float* __restrict__ kpts, const float* __restrict__ line_pts, int B,
                                                                sncal_voter_cfg cfg, const FirstPass* __restrict__ fp,
                                                                sncal_camera* __restrict__ out, int defer_voter) {
    const int frame = (int)blockIdx.x, lane = threadIdx.x & 63;
    if (frame >= B) return;
    float kp[3] = {0.f, 0.f, -1.f};
    if (lane < NPTS) {
        const float* src = kpts + ((size_t)frame * NPTS + lane) * 3;
        kp[0] = src[0]; kp[1] = src[1]; kp[2] = src[2];
    }
    const float* lp = line_pts ? line_pts + (size_t)frame * 90 : nullptr;
    Pts p;
    load_points(kp, p);
    const u64 mask = ov_points(kp, lp, cfg, cfg.algorithm == 0 ? 0.5 : cfg.conf_thresh, p);
    Cam cam;
    cam.tag = SNCAL_CAM_NONE;
    const int st = ov_combine(mask, p, fp[frame].hs, fp[frame].hom, fp[frame].cs, fp[frame].cal, cam);
    if (lane != 0) return;
    if (defer_voter && st != ST_OK) {        // iterative_voter, prediction.py:245-257: left for the voter stage
        store_camera(out + frame, ST_NONE, cam);
        out[frame].status = STATUS_PENDING;
    } else {
        store_camera(out + frame, st, cam);
    }
}

// stand-alone Camera.refine_camera / Camera.solve_pnp on caller-provided 3-D / 2-D matches
__global__ __launch_bounds__(64) void pnp_kernel(const double* __restrict__ Kin, const double* __restrict__ p3,
                                                 const double* __restrict__ p2, const int* __restrict__ npts, int N,
                                                 double* __restrict__ rt, double* __restrict__ rmse, int mode,
                                                 int max_iters, double eps, int sched) {
    const int b = blockIdx.x, lane = threadIdx.x & 63;
    const int n = min(npts[b], min(N, 64));
    const bool in = lane < n;
    double X[3] = {0, 0, 0}, u = 0, v = 0;
    if (in) {
        const double* q = p3 + ((size_t)b * N + lane) * 3;
        X[0] = q[0]; X[1] = q[1]; X[2] = q[2];
        u = p2[((size_t)b * N + lane) * 2]; v = p2[((size_t)b * N + lane) * 2 + 1];
    }
    const u64 mask = n >= 64 ? ~0ull : ((1ull << n) - 1);
    const K4 k{Kin[b * 4 + 0], Kin[b * 4 + 1], Kin[b * 4 + 2], Kin[b * 4 + 3]};
    double R[9], t[3], pos[3];
    for (int i = 0; i < 9; ++i) R[i] = rt[b * 12 + i];
    for (int i = 0; i < 3; ++i) pos[i] = rt[b * 12 + 9 + i];
    bool ok = true;
    if (mode == 0) {
        for (int i = 0; i < 3; ++i) t[i] = -(R[i * 3] * pos[0] + R[i * 3 + 1] * pos[1] + R[i * 3 + 2] * pos[2]);
        if (sched == SCHED_OPENCV) lm_solver_pose_auto(mask, R, t, k, X, u, v, max_iters, eps);
        else refine_pose_lm(mask, R, t, k, X, u, v, max_iters, eps);
    } else {
        // plane membership for the minimal solver = points with z == 0 (ids outside top_gates)
        const u64 gm = __ballot(in && X[2] == 0.0);
        ok = pnp_ransac(sched, mask, gm, k, X, u, v, R, t);
    }
    if (lane == 0 && ok) {
        for (int i = 0; i < 9; ++i) rt[b * 12 + i] = R[i];
        rt[b * 12 + 9] = -(R[0] * t[0] + R[3] * t[1] + R[6] * t[2]);
        rt[b * 12 + 10] = -(R[1] * t[0] + R[4] * t[1] + R[7] * t[2]);
        rt[b * 12 + 11] = -(R[2] * t[0] + R[5] * t[1] + R[8] * t[2]);
    }
    if (rmse) {
        double z;
        const double e2 = reproj_e2(R, t, k, X, u, v, &z);
        const double m = wsum(in ? sqrt(e2) : 0.0) / (double)(n > 0 ? n : 1);
        if (lane == 0) rmse[b] = ok ? m : -1.0;
    }
}

// ---- host: pitch template upload ----------------------------------------------------------------------
void tangent_points(const double* c, double r, const double* p, double* a, double* b) {   // ellipse.py:20-33
    const double hyp = std::sqrt((p[0] - c[0]) * (p[0] - c[0]) + (p[1] - c[1]) * (p[1] - c[1]));
    const double th = std::acos(r / hyp), d = std::atan2(p[1] - c[1], p[0] - c[0]);
    a[0] = c[0] + r * std::cos(d + th); a[1] = c[1] + r * std::sin(d + th); a[2] = 0;
    b[0] = c[0] + r * std::cos(d - th); b[1] = c[1] + r * std::sin(d - th); b[2] = 0;
}

void build_pitch(double (*P)[3]) {   // soccerpitch.py:109-263 + ellipse.py:16-92, ids per ellipse.py:99-157
    const double hl = 52.5, hw = 34.0, PL = 16.5, PW = 40.32, GL = 5.5, GW = 18.32, PM = 11.0, Rr = 9.15, gy = 3.66, GH = 2.44;
    auto set = [&](int i, double x, double y, double z) { P[i][0] = x; P[i][1] = y; P[i][2] = z; };
    set(0, -hl, gy, -GH); set(1, -hl, -gy, -GH); set(2, -hl, gy, 0); set(3, -hl, -gy, 0);
    set(4, -hl + GL, GW / 2, 0); set(5, -hl + GL, -GW / 2, 0); set(6, -hl, GW / 2, 0); set(7, -hl, -GW / 2, 0);
    set(8, -hl + PL, PW / 2, 0); set(9, -hl + PL, -PW / 2, 0); set(10, -hl, PW / 2, 0); set(11, -hl, -PW / 2, 0);
    set(12, -hl, hw, 0); set(13, -hl, -hw, 0); set(14, 0, hw, 0); set(15, 0, -hw, 0);
    set(16, hl - PL, PW / 2, 0); set(17, hl - PL, -PW / 2, 0); set(18, hl, PW / 2, 0); set(19, hl, -PW / 2, 0);
    set(20, hl - GL, GW / 2, 0); set(21, hl - GL, -GW / 2, 0); set(22, hl, GW / 2, 0); set(23, hl, -GW / 2, 0);
    set(24, hl, -gy, -GH); set(25, hl, gy, -GH); set(26, hl, -gy, 0); set(27, hl, gy, 0);
    set(28, hl, hw, 0); set(29, hl, -hw, 0);
    const double c0[2] = {0, 0};
    double a[3], b[3];
    tangent_points(c0, Rr, P[15], a, b);
    set(30, a[0], a[1], 0); set(31, b[0], b[1], 0);
    tangent_points(c0, Rr, P[14], a, b);
    set(32, b[0], b[1], 0); set(33, a[0], a[1], 0);
    const double s = std::sqrt(2.0) * Rr / 2;
    set(34, s, -s, 0); set(35, -s, -s, 0); set(36, s, s, 0); set(37, -s, s, 0);
    set(38, Rr, 0, 0); set(39, -Rr, 0, 0); set(40, 0, -Rr, 0); set(41, 0, Rr, 0); set(42, 0, 0, 0);
    const double lpm[2] = {-hl + PM, 0}, rpm[2] = {hl - PM, 0};
    const double dx = PL - PM, ay = std::sqrt(Rr * Rr - dx * dx);
    set(43, lpm[0] + Rr, 0, 0); set(44, -hl + PL, ay, 0); set(45, -hl + PL, -ay, 0);
    tangent_points(lpm, Rr, P[9], a, b); set(46, a[0], a[1], 0);
    tangent_points(lpm, Rr, P[8], a, b); set(47, b[0], b[1], 0);
    set(48, lpm[0], 0, 0); set(49, P[8][0], 0, 0);
    set(50, rpm[0] - Rr, 0, 0); set(51, hl - PL, ay, 0); set(52, hl - PL, -ay, 0);
    tangent_points(rpm, Rr, P[17], a, b); set(53, b[0], b[1], 0);
    tangent_points(rpm, Rr, P[16], a, b); set(54, a[0], a[1], 0);
    set(55, rpm[0], 0, 0); set(56, P[16][0], 0, 0);
}

int ensure_pitch_uploaded() {
    static std::once_flag once;
    static int rc = SNCAL_OK;
    // constant memory is per device: upload for the current device every time it changes
    static thread_local int last_dev = -1;
    int dev = 0;
    SNCAL_CHECK_HIP(hipGetDevice(&dev));
    if (dev == last_dev) return rc;
    double P[NPTS][3], P32[NPTS][3];
    build_pitch(P);
    for (int i = 0; i < NPTS; ++i)
        for (int j = 0; j < 3; ++j) P32[i][j] = (double)(float)P[i][j];
    SNCAL_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(c_P64), P, sizeof(P)));
    SNCAL_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(c_P32), P32, sizeof(P32)));
    last_dev = dev;
    (void)once;
    return SNCAL_OK;
}

}  // namespace

// Scratch of the first pass (B x FirstPass) and of the voter's task stage (B x thresholds x VoterShared, ~0.8 KB each): one buffer per (device, stream),
// grown on demand and kept for the life of the process.  Calls on one stream are ordered, so the buffer is free again when the next
// call on that stream reaches its task stage.  Rounds 3-4 took it from hipMallocAsync / hipFreeAsync per call: with the solves of
// several batches on several streams (pipeline.py, round 5) an allocation that wants to reuse a block freed on ANOTHER stream made
// the HOST wait for that stream's solve -- the next forward was enqueued 164 ms late, measured (tools/dev/trace_queues.py).
namespace {
struct ScratchBuf { void* p = nullptr; size_t bytes = 0; };
std::mutex g_scratch_mu;
std::map<std::pair<int, hipStream_t>, ScratchBuf> g_scratch;
}
static int voter_scratch(hipStream_t st, size_t bytes, void** out) {
    int dev = 0;
    SNCAL_CHECK_HIP(hipGetDevice(&dev));
    void* old = nullptr;
    {   std::lock_guard<std::mutex> lock(g_scratch_mu);
        ScratchBuf& b = g_scratch[{dev, st}];
        if (b.bytes >= bytes) { *out = b.p; return SNCAL_OK; }
        old = b.p;                      // growth: the entry is taken out under the lock, the wait for the stream's solves happens outside it
        b.p = nullptr; b.bytes = 0;
    }
    if (old) { SNCAL_CHECK_HIP(hipStreamSynchronize(st)); (void)hipFree(old); }
    const size_t want = std::max(bytes, (size_t)1 << 20);
    void* p = nullptr;
    SNCAL_CHECK_HIP(hipMalloc(&p, want));
    {   std::lock_guard<std::mutex> lock(g_scratch_mu);
        ScratchBuf& b = g_scratch[{dev, st}];
        if (b.p) {                      // another thread grew the same (device, stream) entry meanwhile: keep the larger block
            if (b.bytes >= want) { (void)hipFree(p); *out = b.p; return SNCAL_OK; }
            void* q = b.p; b.p = nullptr;
            (void)hipStreamSynchronize(st); (void)hipFree(q);
        }
        b.p = p; b.bytes = want;
    }
    *out = p;
    return SNCAL_OK;
}

namespace sncal {
// sncal_stream_destroy / sncal_shutdown: the convenience form of sncal_calibrate keeps one scratch block per (device, stream); a stream
// that is destroyed takes its block with it (a later stream at the same address must not inherit a block the old stream's kernels may
// still be using: the release synchronises the stream first).  st == nullptr with all == true: every block of every device.
int release_solve_scratch(hipStream_t st, bool all) {
    std::vector<std::pair<std::pair<int, hipStream_t>, void*>> victims;
    {   std::lock_guard<std::mutex> lock(g_scratch_mu);
        for (auto it = g_scratch.begin(); it != g_scratch.end();) {
            if (all || it->first.second == st) { if (it->second.p) victims.push_back({it->first, it->second.p}); it = g_scratch.erase(it); }
            else ++it;
        }
    }
    int cur = 0;
    (void)hipGetDevice(&cur);
    for (auto& v : victims) {
        (void)hipSetDevice(v.first.first);
        (void)hipStreamSynchronize(v.first.second);
        (void)hipFree(v.second);
    }
    (void)hipSetDevice(cur);
    return SNCAL_OK;
}
}  // namespace sncal

static size_t calibrate_fp_bytes(int B) { return ((size_t)B * sizeof(FirstPass) + 255) & ~(size_t)255; }
static size_t calibrate_ws_bytes(int B, const sncal_voter_cfg* cfg) {
    return calibrate_fp_bytes(B) + (size_t)B * std::max(cfg->n_conf_threshs, 1) * sizeof(VoterShared);
}

extern "C" int sncal_calibrate_workspace(int B, const sncal_voter_cfg* cfg, size_t* bytes) {
    SNCAL_CHECK_ARG(B >= 0 && cfg && bytes, "sncal_calibrate_workspace: bad arguments");
    SNCAL_CHECK_ARG(cfg->n_conf_threshs >= 0 && cfg->n_conf_threshs <= SNCAL_MAX_CONF_THRESHS, "sncal_calibrate_workspace: n_conf_threshs");
    *bytes = std::max(calibrate_ws_bytes(B, cfg), (size_t)256);
    return SNCAL_OK;
}

static int calibrate_impl(const float* d_kpts, const float* d_line_pts, int B, const sncal_voter_cfg* cfg,
                          sncal_camera* d_out, void* d_ws, size_t ws_bytes, bool own_ws, void* stream) {
    SNCAL_CHECK_ARG(B >= 0 && cfg, "sncal_calibrate: bad arguments");
    if (B == 0) return SNCAL_OK;
    SNCAL_CHECK_ARG(d_kpts && d_out, "sncal_calibrate: null pointer");
    SNCAL_CHECK_ARG(cfg->algorithm >= 0 && cfg->algorithm <= 4, "sncal_calibrate: algorithm %d", cfg->algorithm);
    SNCAL_CHECK_ARG(cfg->n_conf_threshs >= 0 && cfg->n_conf_threshs <= SNCAL_MAX_CONF_THRESHS, "sncal_calibrate: n_conf_threshs");
    SNCAL_CHECK_ARG(cfg->img_w > 1 && cfg->img_h > 1, "sncal_calibrate: image size");
    const int rc = ensure_pitch_uploaded();
    if (rc) return rc;
    // iterative_voter: frames whose first pass (original_voter) fails fall through to the voter at up to three thresholds,
    // 4x the work of the common case; in one kernel they were stragglers that set its duration (8.5 ms for 64 frames of
    // which 61 were done after 2.7 ms).  They are finished by a second launch that spreads the voter over four waves.
    static const bool split = !(getenv("SNCAL_SOLVE_SPLIT") && atoi(getenv("SNCAL_SOLVE_SPLIT")) == 0);      // tuning aid
    const int defer = (cfg->algorithm == 0 && split && cfg->n_conf_threshs > 0) ? 1 : 0;
    hipStream_t st = sncal::as_stream(stream);
    // Every workgroup of the default path is ONE wavefront (first_pass_task_kernel above says why); SNCAL_SOLVE_WAVE_WGS=0 (tuning
    // aid / A-B reference, also what the byte-identity test compares with): round 4's paired 256-thread calibrate_kernel
    static const bool wave_wgs = !(getenv("SNCAL_SOLVE_WAVE_WGS") && atoi(getenv("SNCAL_SOLVE_WAVE_WGS")) == 0);
    const bool paired = cfg->algorithm <= 1;
    // scratch: the first pass's per-frame slots, then the voter's per-(frame, threshold) slots -- the caller's workspace
    // (sncal_calibrate_ws) or the stream's own block (sncal_calibrate)
    const size_t fp_bytes = calibrate_fp_bytes(B);
    char* scratch = reinterpret_cast<char*>(d_ws);
    if (own_ws) {
        const int rcs = voter_scratch(st, calibrate_ws_bytes(B, cfg), reinterpret_cast<void**>(&scratch));
        if (rcs) return rcs;
    } else {
        SNCAL_CHECK_ARG(d_ws && (reinterpret_cast<uintptr_t>(d_ws) & 15) == 0, "sncal_calibrate_ws: workspace pointer (16-byte aligned device memory)");
        if (ws_bytes < calibrate_ws_bytes(B, cfg)) { sncal::set_error("sncal_calibrate_ws: workspace of %zu bytes, %zu needed (sncal_calibrate_workspace)", ws_bytes, calibrate_ws_bytes(B, cfg)); return SNCAL_ERR_WORKSPACE; }
    }
    if (paired && wave_wgs && (cfg->algorithm == 1 || defer)) {
        FirstPass* fp = reinterpret_cast<FirstPass*>(scratch);
        hipLaunchKernelGGL(first_pass_task_kernel, dim3((unsigned)(2 * B)), dim3(64), 0, st, d_kpts, d_line_pts, B, *cfg, fp);
        SNCAL_CHECK_LAUNCH();
        hipLaunchKernelGGL(first_pass_combine_kernel, dim3((unsigned)B), dim3(64), 0, st, d_kpts, d_line_pts, B, *cfg, (const FirstPass*)fp, d_out, defer);
        SNCAL_CHECK_LAUNCH();
    } else if (paired) {
        hipLaunchKernelGGL(calibrate_kernel, dim3((unsigned)((B + 1) / 2)), dim3(256), 0, st, d_kpts, d_line_pts, B, *cfg, d_out, defer);
        SNCAL_CHECK_LAUNCH();
    } else {                                 // voter / opencv_calibration(_multiplane): one wavefront per frame, no coupling between frames
        const int wpw = wave_wgs ? 1 : 4;
        hipLaunchKernelGGL(calibrate_kernel, dim3((unsigned)((B + wpw - 1) / wpw)), dim3(64 * wpw), 0, st, d_kpts, d_line_pts, B, *cfg, d_out, defer);
        SNCAL_CHECK_LAUNCH();
    }
    if (defer) {
        // one wavefront per (frame, threshold, camera) of the pending frames, then the selection in the reference's order.
        // SNCAL_SOLVE_TASKS=0 (tuning aid / A-B reference): the four-wave voter_kernel, thresholds one after the other
        static const bool tasks = !(getenv("SNCAL_SOLVE_TASKS") && atoi(getenv("SNCAL_SOLVE_TASKS")) == 0);
        if (tasks) {
            VoterShared* slots = reinterpret_cast<VoterShared*>(scratch + fp_bytes);
            hipLaunchKernelGGL(voter_task_kernel, dim3((unsigned)(B * cfg->n_conf_threshs * VT_TASKS)), dim3(64), 0, st, d_kpts, d_line_pts, B, *cfg,
                               (const sncal_camera*)d_out, slots);
            SNCAL_CHECK_LAUNCH();
            hipLaunchKernelGGL(voter_select_kernel, dim3((unsigned)((B + 63) / 64)), dim3(64), 0, st, B, *cfg, (const VoterShared*)slots, d_out);
            SNCAL_CHECK_LAUNCH();
        } else {
            hipLaunchKernelGGL(voter_kernel, dim3(B), dim3(256), 0, st, d_kpts, d_line_pts, B, *cfg, d_out);
            SNCAL_CHECK_LAUNCH();
        }
    }
    return SNCAL_OK;
}

extern "C" int sncal_calibrate(const float* d_kpts, const float* d_line_pts, int B, const sncal_voter_cfg* cfg,
                               sncal_camera* d_out, void* stream) {
    return calibrate_impl(d_kpts, d_line_pts, B, cfg, d_out, nullptr, 0, true, stream);
}

extern "C" int sncal_calibrate_ws(const float* d_kpts, const float* d_line_pts, int B, const sncal_voter_cfg* cfg,
                                  sncal_camera* d_out, void* d_ws, size_t ws_bytes, void* stream) {
    return calibrate_impl(d_kpts, d_line_pts, B, cfg, d_out, d_ws, ws_bytes, false, stream);
}

static int env_schedule() {       // SNCAL_SOLVE_SCHEDULE=converged: the two single-camera entries on the run-to-convergence minimisers
    static const int v = (getenv("SNCAL_SOLVE_SCHEDULE") && std::string(getenv("SNCAL_SOLVE_SCHEDULE")) == "converged") ? SCHED_CONVERGED : SCHED_OPENCV;
    return v;
}

static int launch_pnp(const double* d_K, const double* d_pts3d, const double* d_pts2d, const int32_t* d_npts, int B, int N,
                      double* d_rt, double* d_rmse, int mode, int max_iters, double eps, void* stream) {
    SNCAL_CHECK_ARG(B >= 0 && N > 0 && N <= 64, "pnp: need 0 < N <= 64 points per frame (got %d)", N);
    if (B == 0) return SNCAL_OK;
    SNCAL_CHECK_ARG(d_K && d_pts3d && d_pts2d && d_npts && d_rt, "pnp: null pointer");
    hipLaunchKernelGGL(pnp_kernel, dim3(B), dim3(64), 0, sncal::as_stream(stream), d_K, d_pts3d, d_pts2d, d_npts, N, d_rt,
                       d_rmse, mode, max_iters, eps, env_schedule());
    SNCAL_CHECK_LAUNCH();
    return SNCAL_OK;
}

extern "C" int sncal_pnp_refine_lm(const double* d_K, const double* d_pts3d, const double* d_pts2d, const int32_t* d_npts,
                                   int B, int N, double* d_rt, double* d_rmse, int max_iters, double eps, void* stream) {
    const bool cv = env_schedule() == SCHED_OPENCV;      // defaults = the criteria camera.py:116-117 passes: (20000, 1e-5)
    return launch_pnp(d_K, d_pts3d, d_pts2d, d_npts, B, N, d_rt, d_rmse, 0, max_iters > 0 ? max_iters : (cv ? 20000 : 100),
                      eps > 0 ? eps : (cv ? 1e-5 : 1e-10), stream);
}

extern "C" int sncal_solve_pnp(const double* d_K, const double* d_pts3d, const double* d_pts2d, const int32_t* d_npts,
                               int B, int N, double* d_rt, void* stream) {
    return launch_pnp(d_K, d_pts3d, d_pts2d, d_npts, B, N, d_rt, nullptr, 1, 20, 1e-10, stream);
}
