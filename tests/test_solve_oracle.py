"""CPU: the camera-solve oracle (oracle/solve.py).  OpenCV parity is UNPINNED (cv2 is not available and the
reference holds no fixture for this stage), so the oracle is pinned by
  (1) known-camera recovery on synthetic keypoints,
  (2) an independent scipy.optimize.least_squares cross-check of each minimiser,
  (3) the committed fixture tests/golden/solve_cameras.npz (regression pin of the build's own numbers).
"""
import os

import numpy as np
import pytest
from scipy.optimize import least_squares

from oracle import camera_math as cm
from oracle import solve, synth
from oracle.pitch import GROUND, pitch_points

P = pitch_points()


def _clean_frame(seed, sigma=0.5, min_visible=12):
    rng = np.random.Generator(np.random.PCG64(1000 + seed))
    while True:
        cam = synth.sample_camera(rng)
        uv, vis = synth.project_template(cam)
        if vis.sum() >= min_visible:
            break
    ids = list(np.nonzero(vis)[0])
    obs = uv[ids] + rng.normal(0, sigma, (len(ids), 2))
    return cam, ids, obs


def test_sampler_is_deterministic_and_distinct():
    for n in (4, 5, 9, 53):
        for h in range(128):
            idx = solve.sample4(h, n)
            assert idx is None or (len(set(idx)) == 4 and max(idx) < n)
    assert solve.sample4(0, 8) == solve.sample4(0, 8)
    assert solve._mix(1, 2) == 0x111f7b3bca172522 and solve.sample4(5, 9) == [5, 0, 7, 2]      # known answers (shared with solve.hip)


def test_homography_exact_and_ransac_rejects_outliers():
    cam, ids, _ = _clean_frame(0, sigma=0.0)
    g = [i for i in ids if i in GROUND]
    R, t = cam['rotation'], -cam['rotation'] @ cam['position']
    K = np.array([[cam['f'], 0, 480.], [0, cam['f'], 270.], [0, 0, 1.]])
    Ht = K @ np.column_stack([R[:, 0], R[:, 1], t])
    Ht /= Ht[2, 2]
    uv = solve._apply_h(Ht, P[g, :2])
    H = solve.homography_ransac(P[g, :2], uv, 10.0)
    assert np.abs(H - Ht).max() < 1e-6 * np.abs(Ht).max()
    uv_bad = uv.copy()
    uv_bad[1] += (150.0, -90.0)                                    # one gross outlier
    H2 = solve.homography_ransac(P[g, :2], uv_bad, 10.0)
    err = np.linalg.norm(solve._apply_h(H2, P[g, :2]) - uv, axis=1)
    assert np.delete(err, 1).max() < 1e-4
    assert solve.homography_ransac(P[g[:3], :2], uv[:3], 10.0) is None          # < 4 points
    ok, fx, fy = cm.k_from_plane_homography(Ht)
    assert ok and abs(fx - cam['f']) < 1e-6 * cam['f']


@pytest.mark.parametrize('seed', range(4))
def test_refine_pose_lm_matches_scipy_minimum(seed):
    cam, ids, obs = _clean_frame(seed, sigma=1.0)
    K4 = (cam['f'], cam['f'], 479.5, 269.5)
    R0 = solve.exp_so3(np.array([0.01, -0.02, 0.015])) @ cam['rotation']
    t0 = -R0 @ (cam['position'] + np.array([0.5, -0.4, 0.3]))
    R, t = solve.refine_pose_lm(R0, t0, K4, P[ids], obs)

    def res(x):
        Rx = solve.exp_so3(x[:3]) @ R0
        p, _ = solve.project(Rx, x[3:], K4, P[ids])
        return (p - obs).ravel()
    sp = least_squares(res, np.r_[0, 0, 0, t0], method='lm', xtol=1e-14, ftol=1e-14, gtol=1e-14)
    c_mine = float(((solve.project(R, t, K4, P[ids])[0] - obs) ** 2).sum())
    assert abs(c_mine - 2 * sp.cost) <= 1e-8 * max(2 * sp.cost, 1e-12)          # same minimum (to 1e-8)
    assert np.abs(R @ R.T - np.eye(3)).max() < 1e-12


@pytest.mark.parametrize('seed', range(3))
def test_calibrate_planes_matches_scipy_joint_minimum(seed):
    cam, ids, obs = _clean_frame(10 + seed, sigma=0.7, min_visible=14)
    g = [k for k, i in enumerate(ids) if i in GROUND]
    view = (solve.P32[[ids[k] for k in g]], obs[g])
    f, cx, cy, R0, t0 = solve.calibrate_planes([view], [1], (960, 540))
    assert (cx, cy) == (479.5, 269.5)                                             # quirk Q3

    def res(x):
        Rx = solve.exp_so3(x[1:4]) @ R0
        p, _ = solve.project(Rx, x[4:7], (x[0], x[0], cx, cy), view[0])
        return (p - view[1]).ravel()
    sp = least_squares(res, np.r_[f * 1.02, 0, 0, 0, t0], method='lm', xtol=1e-14, ftol=1e-14, gtol=1e-14)
    c_mine = float((res(np.r_[f, 0, 0, 0, t0]) ** 2).sum())
    assert abs(c_mine - 2 * sp.cost) <= 1e-7 * max(2 * sp.cost, 1e-12)
    assert abs(f - sp.x[0]) < 1e-4 * f
    assert abs(f - cam['f']) < 0.05 * cam['f']


def test_known_camera_recovery_all_algorithms():
    """Noise-limited recovery: rmse at the noise floor, focal length and position close to the truth."""
    for alg in ('iterative_voter', 'voter', 'original_voter', 'opencv_calibration', 'opencv_calibration_multiplane'):
        oc = solve.CameraCreatorOracle(algorithm=alg)
        n_ok = 0
        for seed in range(6):
            kp, cam = synth.synth_keypoints(500 + seed, sigma_px=0.7, outlier_frac=0.0, min_visible=14)
            c = oc(kp, None)
            assert c is not None, (alg, seed)
            assert c.rmse < 2.0, (alg, seed, c.rmse)
            if abs(c.xfocal_length - cam['f']) < 0.05 * cam['f'] and np.linalg.norm(c.position - cam['position']) < 3.0:
                n_ok += 1
        assert n_ok >= 5, (alg, n_ok)


def test_too_few_points_returns_none_and_never_raises():
    oc = solve.CameraCreatorOracle()
    kp = np.zeros((57, 3), dtype=np.float32)
    assert oc(kp, None) is None
    kp[[4, 5, 8]] = [[100, 100, 0.9], [200, 120, 0.9], [150, 300, 0.9]]
    assert oc(kp, None) is None
    kp[:] = [10.0, 10.0, 0.9]                      # 57 coincident points: degenerate everything
    assert oc(kp, None) is None


def test_view_duplication_quirk_q1_weights():
    ids = [4, 5, 8, 9, 16, 17, 20, 21, 0, 1, 2, 3, 6, 7]
    uv = np.zeros((len(ids), 2))
    views, weights = solve._views_from(ids, uv, 6, duplicate=True)
    # ground ids list = [2,3,4,...]: first detected is id 2 at index 0 -> 54 copies; goal_left [0,1,2,3,6,7,..] -> 10
    assert weights == [54, 10] and [len(v[0]) for v in views] == [12, 6]
    views, weights = solve._views_from(ids, uv, 6, duplicate=False)
    assert weights == [1, 1]
    assert np.all(views[1][0][:, 2] == 0)          # goal plane expressed as a z=0 view (swap_z_y)


def test_solve_fixture_regression(gold_dir):
    g = np.load(os.path.join(gold_dir, 'solve_cameras.npz'))
    oc = solve.CameraCreatorOracle()
    for i, seed in enumerate(g['seeds']):
        kp, _ = synth.synth_keypoints(int(seed), sigma_px=1.0)
        assert np.array_equal(kp, g['kpts'][i])
        c = oc(kp, None)
        if g['status'][i] == 0:
            assert c is None
        else:
            assert c is not None and abs(c.rmse - g['rmse'][i]) <= 1e-6 * g['rmse'][i]
            assert abs(c.xfocal_length - g['f'][i]) <= 1e-6 * g['f'][i]
