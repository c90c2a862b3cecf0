#!/usr/bin/env python3
"""Turn "OpenCV parity unpinned" into a committed fixture -- on any machine that has opencv-python==4.7.0.72 (the reference's pin,
requirements.txt:5; cv2 is neither vendored in the reference nor installable in the build container, so this script has never
run there).

    pip install opencv-python==4.7.0.72 numpy
    python tools/make_solve_cv2_fixture.py            ->  tests/golden/solve_cv2.npz
    python -m pytest tests/test_solve_cv2_optional.py

For seeded synthetic frames (oracle/synth.py, the SURVEY 8d recipe) it calls cv2 exactly where and how the reference does and stores
inputs and outputs of every call:
    findHomography(ground pts, img pts, cv2.RANSAC, 10)                     src/datatools/ellipse.py:497, prediction.py:497-500
    calibrateCamera(views, img pts, (960,540), None, None, flags=FIX_*)      prediction.py:398-408 (one view per plane)
                                                                             prediction.py:614-623 (duplicated views, quirk Q1)
    solvePnPRansac(obj, img, K, None)  -> Rodrigues                          baseline/camera.py:100-102
    solvePnPRefineLM(obj, img, K, None, rvec, tvec, (ITER+EPS, 20000, 1e-5)) baseline/camera.py:112-118
The fixture is DATA (inputs + OpenCV's outputs); no reference source is copied.  tests/test_solve_cv2_optional.py compares the
oracle's restatements with it (and with a live cv2 when one is importable)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import solve, synth  # noqa: E402
from oracle.pitch import GROUND  # noqa: E402

SEEDS = list(range(48))


def frame_inputs(seed):
    """What the reference hands to cv2 for one frame at conf_thresh 0.5: ids, float image points, plane views."""
    kp, cam = synth.synth_keypoints(seed, sigma_px=1.0)
    ids = [i for i in range(57) if kp[i, 2] > 0.5]
    uv = kp[ids, :2].astype(np.float64)
    uv32 = uv.astype(np.float32)
    views1, _ = solve._views_from(ids, uv32.astype(np.float64), 6, duplicate=False)
    viewsq, wq = solve._views_from(ids, uv32.astype(np.float64), 6, duplicate=True)
    return kp, ids, uv, views1, viewsq, wq


def cv_flags(cv2):
    f = cv2.CALIB_FIX_PRINCIPAL_POINT | cv2.CALIB_FIX_ASPECT_RATIO | cv2.CALIB_FIX_TANGENT_DIST | cv2.CALIB_FIX_S1_S2_S3_S4 | cv2.CALIB_FIX_TAUX_TAUY
    return f | cv2.CALIB_FIX_K1 | cv2.CALIB_FIX_K2 | cv2.CALIB_FIX_K3 | cv2.CALIB_FIX_K4 | cv2.CALIB_FIX_K5 | cv2.CALIB_FIX_K6


def run_cv2(cv2, seed):
    """dict of cv2's outputs for one seed (NaN-filled where the reference's preconditions do not hold)."""
    kp, ids, uv, views1, viewsq, wq = frame_inputs(seed)
    nan = np.full
    out = {'kp': kp, 'H': nan((3, 3), np.nan), 'cal1': nan(10, np.nan), 'calq': nan(10, np.nan), 'pnp': nan(6, np.nan), 'refine': nan(6, np.nan)}
    g = [k for k, i in enumerate(ids) if i in GROUND]
    if len(g) >= 4:
        H, _ = cv2.findHomography(solve.P32[[ids[k] for k in g], :2].astype(np.float32), uv[g].astype(np.float32), cv2.RANSAC, 10.0)
        if H is not None:
            out['H'] = H
    K = None
    for key, views, weights in (('cal1', views1, [1] * len(views1)), ('calq', viewsq, wq)):
        if not views or sum(w * len(v[1]) for v, w in zip(views, weights)) <= 6:
            continue
        obj = [np.asarray(v[0], dtype=np.float32) for v, w in zip(views, weights) for _ in range(w)]
        img = [np.asarray(v[1], dtype=np.float32) for v, w in zip(views, weights) for _ in range(w)]
        try:
            rms, mtx, dist, rv, tv = cv2.calibrateCamera(obj, img, (960, 540), None, None, flags=cv_flags(cv2))
        except cv2.error:
            continue
        out[key] = np.r_[rms, mtx[0, 0], mtx[0, 2], mtx[1, 2], np.asarray(rv[0]).ravel(), np.asarray(tv[0]).ravel()]      # rms f cx cy rvec0 tvec0
        if key == 'calq':
            K = mtx
    if K is not None and len(ids) >= 4:
        obj = solve.P64[ids]
        try:
            ok, rvec, t, inl = cv2.solvePnPRansac(obj, uv, K, None)
        except cv2.error:
            ok = False
        if ok:
            out['pnp'] = np.r_[rvec.ravel(), t.ravel()]
            rv2, t2 = cv2.solvePnPRefineLM(obj, uv, K, None, rvec.copy(), t.copy(),
                                           (cv2.TERM_CRITERIA_MAX_ITER + cv2.TERM_CRITERIA_EPS, 20000, 0.00001))
            out['refine'] = np.r_[rv2.ravel(), t2.ravel()]
    return out


def main():
    import cv2
    rows = [run_cv2(cv2, s) for s in SEEDS]
    path = os.path.join(ROOT, 'tests', 'golden', 'solve_cv2.npz')
    np.savez_compressed(path, seeds=np.array(SEEDS), cv2_version=np.array(cv2.__version__),
                        **{k: np.stack([r[k] for r in rows]) for k in rows[0]})
    print('wrote', path, 'with OpenCV', cv2.__version__, '(the reference pins 4.7.0.72)')


if __name__ == '__main__':
    main()
