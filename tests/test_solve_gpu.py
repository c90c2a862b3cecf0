"""GPU: the HIP camera solve (sncal_calibrate / sncal_solve_pnp / sncal_pnp_refine_lm through the C ABI and
the CameraCreator / Camera host mirrors) vs the numpy oracle and the committed fixture.
Tolerance (north star): reprojection error within 1e-4 relative; None-ness identical."""
import json
import os
import pickle

import numpy as np
import pytest
import torch

from oracle import camera_math as cm
from oracle import solve, synth
from oracle.pitch import pitch_points

pytestmark = pytest.mark.gpu
P = pitch_points()
KW = dict(conf_thresh=0.5, conf_threshs=[0.5, 0.35, 0.2], max_rmse=55.0, max_rmse_rel=5.0, min_points=5,
          min_focal_length=10.0, min_points_per_plane=6, min_points_for_refinement=6, reliable_thresh=57)


def _compare(cams, oracle_cams, rtol=1e-4):
    for i, (c, o) in enumerate(zip(cams, oracle_cams)):
        assert (c is None) == (o is None), i
        if o is None:
            continue
        assert abs(c.rmse - o.rmse) <= rtol * o.rmse, (i, c.rmse, o.rmse)
        assert abs(c.xfocal_length - o.xfocal_length) <= 1e-4 * o.xfocal_length
        assert np.linalg.norm(c.position - o.position) <= 1e-3 * max(1.0, np.linalg.norm(o.position))
        assert np.abs(c.rotation @ c.rotation.T - np.eye(3)).max() < 1e-10


@pytest.mark.parametrize('alg', ['iterative_voter', 'voter', 'original_voter', 'opencv_calibration',
                                 'opencv_calibration_multiplane'])
def test_calibrate_matches_oracle(sncal, cuda, alg):
    seeds = range(100, 132)
    kps = np.stack([synth.synth_keypoints(s, sigma_px=1.0)[0] for s in seeds])
    cc = sncal.CameraCreator(sncal.PITCH_POINTS, algorithm=alg, **KW)
    cams = cc.solve_batch(kps)
    oc = solve.CameraCreatorOracle(algorithm=alg)
    _compare(cams, [oc(k, None) for k in kps])


def test_calibrate_matches_committed_fixture(sncal, cuda, gold_dir):
    g = np.load(os.path.join(gold_dir, 'solve_cameras.npz'))
    cc = sncal.CameraCreator(sncal.PITCH_POINTS, algorithm='iterative_voter', **KW)
    cams = cc.solve_batch(g['kpts'])
    for i, c in enumerate(cams):
        assert (c is None) == (g['status'][i] == 0), i
        if c is not None:
            assert abs(c.rmse - g['rmse'][i]) <= 1e-4 * g['rmse'][i]
            assert abs(c.xfocal_length - g['f'][i]) <= 1e-4 * g['f'][i]
            # the solve's rmse equals Camera.projection_rmse evaluated by the host mirror on the same points
            sel = np.nonzero(g['kpts'][i][:, 2] > 0.5)[0]


def test_single_frame_call_never_raises_and_is_picklable(sncal, cuda):
    cc = sncal.CameraCreator(sncal.PITCH_POINTS, algorithm='iterative_voter', **KW)
    cc2 = pickle.loads(pickle.dumps(cc))                                   # shipped to workers in make_submit.py
    kp, cam = synth.synth_keypoints(500, sigma_px=0.7, outlier_frac=0.0, min_visible=14)
    c = cc2(kp, 'frame.jpg')
    assert c is not None and abs(c.xfocal_length - cam['f']) < 0.05 * cam['f']
    js = c.to_json_parameters()
    assert set(js) == {'pan_degrees', 'tilt_degrees', 'roll_degrees', 'position_meters', 'x_focal_length',
                       'y_focal_length', 'principal_point', 'radial_distortion', 'tangential_distortion',
                       'thin_prism_distortion'}
    json.dumps(js)
    assert js['principal_point'] == [480.0, 270.0] and c.calibration[0, 2] == 479.5          # quirk Q3
    assert cc2(np.zeros((57, 3), dtype=np.float32), None) is None
    assert cc2(np.full((57, 3), np.nan, dtype=np.float32), None) is None
    assert cc2('garbage', None) is None                                    # firewall: prints, returns None


def test_line_points_fill_missing_keypoints(sncal, cuda, tmp_path):
    """prediction.py:105-124 + :356-364: intersections from the lines pickle replace missing keypoints."""
    kp, cam = synth.synth_keypoints(501, sigma_px=0.5, outlier_frac=0.0, min_visible=16)
    vis = np.nonzero(kp[:, 2] > 0.5)[0]
    drop = [i for i in vis if i < 30][:3]
    kp2 = kp.copy()
    kp2[drop, 2] = 0.01
    line_pts = {int(i): (float(kp[i, 0]), float(kp[i, 1])) for i in drop}
    oc = solve.CameraCreatorOracle(lines_data={'a.jpg': line_pts})
    o_with, o_without = oc(kp2, 'a.jpg'), oc(kp2, None)
    cc = sncal.CameraCreator(sncal.PITCH_POINTS, algorithm='iterative_voter', **KW)
    cc.lines_data = {'a.jpg': line_pts}
    c_with, c_without = cc(kp2, 'a.jpg'), cc(kp2, 'other.jpg')
    _compare([c_with, c_without], [o_with, o_without])
    assert abs(c_with.rmse - c_without.rmse) > 1e-9                          # the extra points were really used


def test_camera_mirror_solve_pnp_and_refine(sncal, cuda):
    rng = np.random.Generator(np.random.PCG64(9))
    cam = synth.sample_camera(rng)
    uv, vis = synth.project_template(cam)
    ids = list(np.nonzero(vis)[0])
    obs = uv[ids] + rng.normal(0, 0.8, (len(ids), 2))
    c = sncal.Camera(960, 540)
    c.calibration = np.array([[cam['f'], 0, 480.], [0, cam['f'], 270.], [0, 0, 1.]])
    c.xfocal_length = c.yfocal_length = np.float64(cam['f'])
    matches = [(P[i], tuple(obs[k])) for k, i in enumerate(ids)]
    c.solve_pnp(matches)
    c.refine_camera(matches)
    oc = solve.Cam()
    oc.calibration = c.calibration.copy()
    oc.xfocal_length = oc.yfocal_length = cam['f']
    oc.solve_pnp(ids, obs)
    oc.refine_camera(ids, obs)
    assert np.abs(c.rotation - oc.rotation).max() < 1e-6 and np.abs(c.position - oc.position).max() < 1e-5
    r_host = c.projection_rmse(matches)                                       # host mirror of S8
    assert abs(r_host - oc.projection_rmse(ids, obs)) < 1e-7
    assert np.linalg.norm(c.position - cam['position']) < 2.0
    # JSON round trip keeps the projection (consumer contract H2)
    c2 = sncal.Camera(960, 540)
    c2.from_json_parameters(json.loads(json.dumps(c.to_json_parameters())))
    assert np.abs(c2.project_point(P[ids[0]]) - c.project_point(P[ids[0]])).max() < 1e-6


def test_batch_equals_single_frames(sncal, cuda):
    kps = np.stack([synth.synth_keypoints(s, sigma_px=1.0)[0] for s in range(300, 340)])
    cc = sncal.CameraCreator(sncal.PITCH_POINTS, algorithm='iterative_voter', **KW)
    batch = cc.solve_batch(kps)
    for i in (0, 7, 21, 39):
        one = cc(kps[i], None)
        assert (one is None) == (batch[i] is None)
        if one is not None:
            assert one.rmse == batch[i].rmse and np.array_equal(one.rotation, batch[i].rotation)
