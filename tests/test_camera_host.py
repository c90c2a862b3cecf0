"""Product host arithmetic of sncal_amd.Camera against the reference capture tests/golden/camera.npz
(tools/make_golden.py gen_camera: the imported baseline.camera.Camera run on sampled cameras).  CPU only."""
import json
import os

import numpy as np

import sncal_amd
from sncal_amd.pitch import PITCH_POINTS, INTERSECTON_TO_PITCH_POINTS


def _pitch_array():
    return np.stack([np.asarray(PITCH_POINTS[INTERSECTON_TO_PITCH_POINTS[i]], dtype=np.float64) for i in range(57)])


def _camera(g, i):
    c = sncal_amd.Camera(960, 540)
    c.position = g[f'{i}.position'].copy()
    c.rotation = g[f'{i}.rotation'].copy()
    f = np.float64(g[f'{i}.f'])
    c.xfocal_length = c.yfocal_length = f
    c.calibration = np.array([[f, 0, 480.], [0, f, 270.], [0, 0, 1.]])
    return c


def test_projection_rmse_and_json_match_the_reference_capture(gold_dir):
    g = np.load(os.path.join(gold_dir, 'camera.npz'), allow_pickle=False)
    P = _pitch_array()
    for i in range(int(g['n'])):
        c = _camera(g, i)
        block = c.project_points(P)
        assert np.abs(block - g[f'{i}.proj']).max() < 1e-9
        rows = np.stack([c.project_point(p) for p in P])
        assert np.array_equal(rows, block)                       # one-row form == block form, bit for bit
        ids = g[f'{i}.obs_ids']
        mp = [(P[k], tuple(o)) for k, o in zip(ids, g[f'{i}.obs'])]
        assert abs(c.projection_rmse(mp) - float(g[f'{i}.rmse'])) < 1e-9
        js = json.loads(str(g[f'{i}.json']))
        mine = c.to_json_parameters()
        assert list(mine.keys()) == list(js.keys())              # key order is part of the file format
        for k in ('pan_degrees', 'tilt_degrees', 'roll_degrees'):
            assert abs(js[k] - mine[k]) < 1e-10
        for k in ('position_meters', 'principal_point', 'radial_distortion', 'tangential_distortion', 'thin_prism_distortion'):
            assert np.allclose(js[k], mine[k], rtol=0, atol=1e-12)
        c2 = sncal_amd.Camera(960, 540)
        c2.from_json_parameters(js)
        assert np.abs(c2.rotation - g[f'{i}.rot_from_json']).max() < 1e-12
        assert np.abs(c2.position - c.position).max() < 1e-12 and c2.xfocal_length == js['x_focal_length']
        assert np.array_equal(c2.calibration, np.array([[js['x_focal_length'], 0, 480.], [0, js['y_focal_length'], 270.], [0, 0, 1.]]))


def test_intrinsics_and_pose_from_plane_homography_match_the_reference_capture(gold_dir):
    g = np.load(os.path.join(gold_dir, 'camera.npz'), allow_pickle=False)
    for i in range(int(g['n'])):
        c = sncal_amd.Camera(960, 540)
        ok, K = c.estimate_calibration_matrix_from_plane_homography(g[f'{i}.H'])
        assert ok == bool(g[f'{i}.k_ok'])
        if ok:
            assert abs(K[0, 0] - float(g[f'{i}.k_fx'])) < 1e-6 * K[0, 0] and abs(K[1, 1] - float(g[f'{i}.k_fy'])) < 1e-6 * K[1, 1]
            assert c.principal_point == (480.0, 270.0) and c.calibration[0, 2] == 480.0 and c.calibration[1, 2] == 270.0
        c2 = sncal_amd.Camera(960, 540)
        assert bool(c2.from_homography(g[f'{i}.H'])) == bool(g[f'{i}.fh_ok'])
        if bool(g[f'{i}.fh_ok']):
            assert np.abs(c2.rotation - g[f'{i}.fh_rot']).max() < 1e-6
            assert np.abs(c2.position - g[f'{i}.fh_pos']).max() < 1e-5 * max(1.0, np.abs(g[f'{i}.fh_pos']).max())
            assert abs(c2.xfocal_length - float(g[f'{i}.fh_fx'])) < 1e-6 * float(g[f'{i}.fh_fx'])


def test_degenerate_homographies_are_refused_not_raised():
    c = sncal_amd.Camera(960, 540)
    assert c.estimate_calibration_matrix_from_plane_homography(np.eye(3))[0] is False       # fronto-parallel: e == 0 / omega singular
    assert c.from_homography(np.zeros((3, 3))) is False
    # behind-camera point -> zeros(3), the reference's convention (camera.py:257-258)
    c.position = np.array([0., 0., 10.])
    assert np.array_equal(c.project_point(np.array([0., 0., 0.])), np.zeros(3))


def test_pan_tilt_roll_round_trip_and_branch_choice():
    from sncal_amd.camera import pan_tilt_roll_to_orientation, rotation_matrix_to_pan_tilt_roll
    rng = np.random.default_rng(0)
    for _ in range(50):
        pan, tilt, roll = rng.uniform(-3, 3), rng.uniform(0.05, 3.0), rng.uniform(-1.5, 1.5)
        R = pan_tilt_roll_to_orientation(pan, tilt, roll).T
        p2, t2, r2 = rotation_matrix_to_pan_tilt_roll(R)
        assert np.abs(pan_tilt_roll_to_orientation(p2, t2, r2).T - R).max() < 1e-12
        assert abs(r2) <= abs(roll) + 1e-12
