#!/usr/bin/env python3
"""Generate tests/golden/* by IMPORTING the reference in the build container.

Runs only where /root/reference exists (never on the GPU box, never from tests/).  It installs four
empty stub modules for dependencies that are absent offline (cv2, torchvision, ellipse, argus), imports the
reference's own modules read-only, feeds them build-owned deterministic inputs (oracle/synth.py,
oracle/hrnet_ref.py::seeded_*), and stores small arrays of inputs -> expected outputs.  Nothing from
the reference's source text is stored: fixtures are data.

While it runs it also asserts that the oracle/ restatements agree with the imported reference, i.e.
it is the "pin" of the oracle.  Usage:  python tools/make_golden.py
"""
import io
import json
import os
import sys
import types
import contextlib

import numpy as np
import torch
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = '/root/reference'
GOLD = os.path.join(ROOT, 'tests', 'golden')
sys.path.insert(0, ROOT)


def install_stubs():
    cv2 = types.ModuleType('cv2')
    cv2.RANSAC = 8

    def _nocv(*a, **k):
        raise RuntimeError('cv2 is not available offline')
    for n in ('findHomography', 'calibrateCamera', 'Rodrigues', 'solvePnPRansac', 'solvePnPRefineLM'):
        setattr(cv2, n, _nocv)
    sys.modules['cv2'] = cv2
    tv = types.ModuleType('torchvision')
    tvt = types.ModuleType('torchvision.transforms')
    tvt.ToTensor = object
    tv.transforms = tvt
    sys.modules['torchvision'] = tv
    sys.modules['torchvision.transforms'] = tvt
    ar = types.ModuleType('argus')
    ar.load_model = None
    ar.Model = type('Model', (), {'__init__': lambda self, params=None: None})
    are = types.ModuleType('argus.engine'); are.State = object
    aru = types.ModuleType('argus.utils'); aru.deep_detach = aru.deep_to = None
    ar.engine, ar.utils = are, aru
    sys.modules.update({'argus': ar, 'argus.engine': are, 'argus.utils': aru})
    el = types.ModuleType('ellipse')
    el.LsqEllipse = object
    sys.modules['ellipse'] = el
    sys.path.insert(0, REF)


class AttrDict(dict):
    """dict with attribute access AND `in` (hrnet.py:306,314 uses both)."""
    def __getattr__(self, k):
        v = self[k]
        return AttrDict(v) if isinstance(v, dict) else v


def ref_cfg(cfg):
    c = {k: v for k, v in cfg.items() if k not in ('head',)}
    if c.get('upscale', 1) == 1:
        c.pop('upscale', None)
    return AttrDict(c)


def checksum(a: np.ndarray) -> float:
    return float(np.sum(a.astype(np.float64)))


def gen_pitch():
    from src.datatools.ellipse import PITCH_POINTS, INTERSECTON_TO_PITCH_POINTS
    from src.models.hrnet import prediction as rp
    from src.datatools.line import LINE_CLS
    from src.datatools.intersections import LINE_INTERSECTIONS
    from oracle import pitch as op, lines as ol
    P = np.stack([PITCH_POINTS[INTERSECTON_TO_PITCH_POINTS[i]] for i in range(57)])
    assert np.abs(P - op.pitch_points()).max() < 1e-12, np.abs(P - op.pitch_points()).max()
    assert rp.top_gates == op.TOP_GATES and rp.keep_points == op.KEEP_POINTS
    assert [i for i in rp.point_sets['groundplane'] if i < 57] == op.GROUND
    assert rp.point_sets['goal_left'] == op.GOAL_LEFT and rp.point_sets['goal_right'] == op.GOAL_RIGHT
    assert [LINE_CLS[i] for i in range(23)] == ol.LINE_CLS
    assert {k: tuple(v) for k, v in LINE_INTERSECTIONS.items()} == ol.LINE_INTERSECTIONS
    np.savez(os.path.join(GOLD, 'pitch.npz'), points=P,
             names=np.array([INTERSECTON_TO_PITCH_POINTS[i] for i in range(57)]),
             keep_points=np.array(rp.keep_points), goal_left=np.array(rp.point_sets['goal_left']),
             goal_right=np.array(rp.point_sets['goal_right']), top_gates=np.array(rp.top_gates),
             line_cls=np.array([LINE_CLS[i] for i in range(23)]),
             line_pairs=np.array([list(LINE_INTERSECTIONS[i]) for i in range(30)]))
    print('pitch ok')


def tie_suite(h, w):
    """Engineered (1,58,h,w) log-heatmap exercising the separable first-occurrence tie rule (H1)."""
    lp = np.full((1, 58, h, w), -20.0, dtype=np.float32)
    lp[0, 0, 10, w - 20] = -0.5; lp[0, 0, h - 5, 7] = -0.5        # equal maxima: x from col 7, y from row 10
    lp[0, 1] = -3.0                                              # all-equal channel -> (0,0)
    lp[0, 2, 3, 5] = -1e-9; lp[0, 2, 2, 9] = -2e-9               # distinct logp that collide at exp()==1.0
    lp[0, 3, h - 1, w - 1] = -0.1                                # last pixel
    lp[0, 4, 0, 0] = -0.25
    lp[0, 5, 5, 5] = -np.inf; lp[0, 5, 6, 6] = -1.0              # -inf is legal input
    lp[0, 6, :, 11] = -0.7                                       # a whole column of equal maxima
    lp[0, 7, 13, :] = -0.7                                       # a whole row of equal maxima
    lp[0, 8, 4, 4] = np.float32(-1.0); lp[0, 8, 3, 8] = np.nextafter(np.float32(-1.0), np.float32(0))
    return lp


def gen_decode():
    from src.models.hrnet.transforms import HRNetPredictionTransform
    from src.models.line.transforms import EHMPredictionTransform
    from oracle import decode as od, synth
    out = {}
    cases = []
    lp, kps = synth.synth_logp(list(range(4)), hw=(68, 120))
    cases.append(('gauss_68x120', lp))
    lp, _ = synth.synth_logp([7, 8], hw=(135, 240), floor=0)      # exact zeros -> -inf
    cases.append(('gauss_135x240_neginf', lp))
    cases.append(('ties_34x60', tie_suite(34, 60)))
    rng = np.random.Generator(np.random.PCG64(5))
    noise = torch.log_softmax(torch.from_numpy((rng.random((2, 58, 34, 60)) * 8).astype(np.float32)), 1).numpy()
    cases.append(('noise_34x60', noise))
    tr = HRNetPredictionTransform((540, 960))
    for name, lp in cases:
        ref = tr(torch.from_numpy(lp)).numpy()
        mine = od.keypoint_decode(lp, (540, 960))
        assert np.array_equal(ref[..., :2], mine[..., :2]), name
        assert np.allclose(ref[..., 2], mine[..., 2], rtol=2e-7, atol=0), name
        out[name + '.in'] = lp if lp.size < 300000 else np.zeros(0, np.float32)
        out[name + '.out'] = ref
    out['seeds.gauss_68x120'] = np.arange(4)
    out['seeds.gauss_135x240_neginf'] = np.array([7, 8])
    np.savez_compressed(os.path.join(GOLD, 'decode_keypoints.npz'), **out)
    # line decode
    out = {}
    rng = np.random.Generator(np.random.PCG64(11))
    B, C, H, W = 2, 23, 34, 60
    heat = np.zeros((B, C, H, W), dtype=np.float32)
    for b in range(B):
        for c in range(C):
            for _ in range(2):
                x0, y0 = rng.uniform(0, W), rng.uniform(0, H)
                amp = rng.uniform(0.1, 1.0)
                xs = np.arange(W)[None, :]; ys = np.arange(H)[:, None]
                heat[b, c] += (amp * np.exp(-((xs - x0) ** 2 + (ys - y0) ** 2) / (2 * 1.5 ** 2))).astype(np.float32)
    heat -= 0.02
    heat[0, 3] = 0.0                       # empty channel
    heat[1, 5] = -1.0                      # all negative -> relu -> zeros
    for sigma, scale in ((3.0, 4.0), (6.0, 4.0)):
        ref = EHMPredictionTransform(scale=scale, sigma=sigma)(torch.from_numpy(heat.copy())).numpy()
        mine = od.line_decode(heat, sigma, scale)
        assert np.array_equal(ref[..., :2], mine[..., :2]), (sigma, np.abs(ref - mine).max())
        assert np.allclose(ref[..., 2], mine[..., 2], rtol=1e-5, atol=1e-7)
        out[f'out_sigma{int(sigma)}'] = ref
    out['heat'] = heat
    np.savez_compressed(os.path.join(GOLD, 'decode_lines.npz'), **out)
    print('decode ok')


def gen_camera():
    from baseline.camera import Camera
    from src.models.hrnet.prediction import good_camera
    from oracle import camera_math as cm, synth, pitch as op
    P = op.pitch_points()
    recs = []
    for seed in range(8):
        rng = np.random.Generator(np.random.PCG64(100 + seed))
        cam = synth.sample_camera(rng)
        c = Camera(960, 540)
        c.position = cam['position'].copy(); c.rotation = cam['rotation'].copy()
        c.xfocal_length = c.yfocal_length = np.float64(cam['f'])   # cv2 / numpy matrices hold float64
        c.calibration = np.array([[cam['f'], 0, 480.], [0, cam['f'], 270.], [0, 0, 1.]])
        proj = np.stack([c.project_point(p) for p in P])
        mine = np.stack([cm.project_point(cam['position'], cam['rotation'], cam['f'], cam['f'], (480., 270.), p) for p in P])
        assert np.abs(proj - mine).max() < 1e-9
        js = c.to_json_parameters()
        mj = cm.to_json(cam['position'], cam['rotation'], cam['f'], cam['f'], (480., 270.))
        for k in ('pan_degrees', 'tilt_degrees', 'roll_degrees'):
            assert abs(js[k] - mj[k]) < 1e-10
        c2 = Camera(960, 540); c2.from_json_parameters(js)
        assert np.abs(c2.rotation - cm.rotation_from_ptr(*np.deg2rad([js['pan_degrees'], js['tilt_degrees'], js['roll_degrees']]))).max() < 1e-12
        vis = proj[:, 2] > 0
        obs = proj[vis, :2] + rng.normal(0, 1.5, (int(vis.sum()), 2))
        mp = [(P[i], tuple(obs[k])) for k, i in enumerate(np.nonzero(vis)[0])]
        rm = c.projection_rmse(mp)
        assert abs(rm - cm.projection_rmse(cam['position'], cam['rotation'], cam['f'], cam['f'], (480., 270.), P[vis], obs)) < 1e-9
        # plane homography world(z=0) -> image and K-from-H
        K = c.calibration; R = c.rotation; t = -R @ c.position
        Hm = K @ np.column_stack([R[:, 0], R[:, 1], t]); Hm /= Hm[2, 2]
        ch = Camera(960, 540)
        ok, Kest = ch.estimate_calibration_matrix_from_plane_homography(Hm)
        ok2, fx2, fy2 = cm.k_from_plane_homography(Hm)
        assert ok == ok2 and (not ok or (abs(Kest[0, 0] - fx2) < 1e-6 * fx2 and abs(Kest[1, 1] - fy2) < 1e-6 * fy2))
        ch2 = Camera(960, 540); okh = ch2.from_homography(Hm)
        recs.append(dict(position=cam['position'], rotation=cam['rotation'], f=cam['f'], proj=proj,
                         json=js, rot_from_json=c2.rotation, obs_ids=np.nonzero(vis)[0], obs=obs, rmse=rm,
                         H=Hm, k_ok=ok, k_fx=Kest[0, 0], k_fy=Kest[1, 1], fh_ok=bool(okh),
                         fh_rot=ch2.rotation, fh_pos=ch2.position, fh_fx=ch2.xfocal_length,
                         good=bool(good_camera(c.calibration, c.position))))
    flat = {}
    for i, r in enumerate(recs):
        for k, v in r.items():
            flat[f'{i}.{k}'] = json.dumps(v) if isinstance(v, dict) else np.asarray(v)
    flat['n'] = np.array(len(recs))
    np.savez(os.path.join(GOLD, 'camera.npz'), **flat)
    print('camera ok')


def gen_lines():
    from src.utils.export_line_result import get_line_data, calculate_slope_intercept
    from src.models.hrnet.prediction import line_eq_intersection, CameraCreator
    from src.datatools.ellipse import PITCH_POINTS
    from oracle import lines as ol
    import pickle
    import tempfile
    g = np.load(os.path.join(GOLD, 'decode_lines.npz'))
    hl = g['out_sigma3'][:1] / np.array([4, 4, 1], dtype=np.float32)   # back to heatmap units
    # The reference pins numpy 1.24.2, whose scalar promotion turns `x2 - x1 + delta` into float64; numpy 2 (installed
    # here) would keep float32.  Peak coordinates are integers, so feeding the reference float64 copies reproduces
    # the pinned arithmetic exactly (every float32 step before the promotion is exact); the oracle gets the float32
    # array and must agree to the last bit.
    lines, points = get_line_data(hl.astype(np.float64), scale=4, prob_thre=0.2)
    ml, mp = ol.get_line_data(hl, scale=4, prob_thre=0.2)
    assert lines.keys() == ml.keys()
    for k in lines:
        assert tuple(float(v) for v in lines[k]) == tuple(float(v) for v in ml[k]), (k, lines[k], ml[k])
    assert calculate_slope_intercept((1., 2.), (1., 2.)) == ol.slope_intercept((1., 2.), (1., 2.)) == (None, None)
    # ingestion through CameraCreator.__init__ (prediction.py:105-124)
    pk = {'img_a.jpg': {'lines': [dict(lines)], 'points': [points]}}
    pk['img_a.jpg']['lines'][0]['Goal left post left'] = (0.3, 11.0)     # key without trailing blank
    pk['img_a.jpg']['lines'][0].pop('Goal left post left ', None)
    with tempfile.NamedTemporaryFile(suffix='.pkl', delete=False) as f:
        pickle.dump(pk, f)
    with contextlib.redirect_stdout(io.StringIO()):
        cc = CameraCreator(PITCH_POINTS, lines_file=f.name)
    os.unlink(f.name)
    got = cc.lines_data.get('img_a.jpg', {})
    mine = ol.lines_to_keypoints(pk['img_a.jpg']['lines'][0])
    assert got.keys() == mine.keys() and all(np.allclose(got[k], mine[k]) for k in got)
    assert line_eq_intersection((1.0, 0.0), (1.00001, 5.0)) is None
    rec = {'lines': {k: [float(v[0]), float(v[1])] for k, v in pk['img_a.jpg']['lines'][0].items()},
           'keypoints': {str(k): [float(v[0]), float(v[1])] for k, v in got.items()},
           'points': {k: [[float(x) for x in p] for p in v] for k, v in points.items()}}
    with open(os.path.join(GOLD, 'lines.json'), 'w') as f:
        json.dump(rec, f, indent=1)
    print('lines ok', len(got), 'intersections')


def gen_evaluator():
    """H2: get_polylines / evaluate_camera_prediction on the Camera JSON contract (numpy only)."""
    from baseline.evaluate_camera import get_polylines, evaluate_camera_prediction
    from oracle import synth, camera_math as cm
    recs = {}
    for seed in range(3):
        rng = np.random.Generator(np.random.PCG64(200 + seed))
        cam = synth.sample_camera(rng)
        js = cm.to_json(cam['position'], cam['rotation'], cam['f'], cam['f'], (480., 270.))
        poly = get_polylines(js, 960, 540, sampling_factor=0.9)
        # perturbed camera as "prediction"
        cam2 = dict(cam); cam2['position'] = cam['position'] + np.array([0.3, -0.2, 0.1])
        js2 = cm.to_json(cam2['position'], cam2['rotation'], cam2['f'] * 1.01, cam2['f'] * 1.01, (480., 270.))
        poly2 = get_polylines(js2, 960, 540, sampling_factor=0.9)
        gt = {k: [v[0], v[-1]] for k, v in poly.items()}
        conf, _, errs = evaluate_camera_prediction(poly2, gt, 5)
        recs[str(seed)] = {'json': js, 'pred_json': js2,
                           'n_classes': len(poly), 'npts': {k: len(v) for k, v in poly.items()},
                           'first': {k: [v[0]['x'], v[0]['y']] for k, v in poly.items()},
                           'confusion': conf.tolist(),
                           'mean_err': {k: float(np.mean(v)) for k, v in errs.items()}}
    with open(os.path.join(GOLD, 'evaluator.json'), 'w') as f:
        json.dump(recs, f, indent=1)
    print('evaluator ok')


def gen_evaluator_batch():
    """N2: the per-frame evaluation of evaluate_camera.py:293-320 (polylines, plain + mirrored confusion, accuracy
    choice) captured from the reference for a batch of predicted cameras and synthetic annotations, plus the sampled
    pitch model (SoccerPitch.sample_field_points(0.9)) -- the oracle must reproduce both exactly."""
    from baseline.evaluate_camera import get_polylines as _get_polylines, evaluate_camera_prediction
    from baseline.evaluate_extremities import mirror_labels
    from baseline.soccerpitch import SoccerPitch
    from baseline.camera import Camera
    from oracle import synth, camera_math as cm, evaluate as oe

    def get_polylines(js, w, h, sampling_factor):
        # Camera.project_point multiplies distort()'s float32 by the focal length: a python float after
        # from_json_parameters, i.e. float64 arithmetic under the reference's pinned numpy 1.24.2 but float32 under
        # the numpy 2 installed here.  np.float64 fields give the pinned arithmetic under both.
        cam = Camera(w, h)
        cam.from_json_parameters(js)
        cam.xfocal_length, cam.yfocal_length = np.float64(cam.xfocal_length), np.float64(cam.yfocal_length)
        cam.principal_point = (np.float64(cam.principal_point[0]), np.float64(cam.principal_point[1]))
        return _get_polylines(cam, w, h, sampling_factor=sampling_factor)
    ref_tab = SoccerPitch().sample_field_points(0.9)
    assert list(ref_tab) == oe.CLASSES and oe.SYMMETRIC == SoccerPitch.symetric_classes
    pts, start = oe.field_table()
    assert np.array_equal(pts, np.array([p for c in oe.CLASSES for p in ref_tab[c]]))
    out = {'field_points': pts, 'class_start': start, 'classes': np.array(oe.CLASSES)}
    n = 0
    for seed in range(10):
        rng = np.random.Generator(np.random.PCG64(900 + seed))
        cam = synth.sample_camera(rng)
        js_true = cm.to_json(cam['position'], cam['rotation'], cam['f'], cam['f'], (480., 270.))
        lvl = 1 if seed == 7 else seed % 4
        noise_pos = rng.normal(0, [0.0, 0.15, 0.6, 2.0][lvl], size=3)
        fpred = cam['f'] * (1 + rng.normal(0, [0.0, 0.004, 0.02, 0.08][lvl]))
        js_pred = cm.to_json(cam['position'] + noise_pos, cam['rotation'], fpred, fpred, (480., 270.))
        poly_true = get_polylines(js_true, 960, 540, sampling_factor=0.9)
        gt = {}
        for k, v in poly_true.items():
            if rng.uniform() < 0.15:
                continue                                             # annotator missed this class -> false positive
            step = int(rng.integers(2, 9))
            sel = v[::step] if len(v) > 2 else v
            gt[k] = [{'x': p['x'] + rng.normal(0, 0.7), 'y': p['y'] + rng.normal(0, 0.7)} for p in sel][:12]
        if seed % 3 == 0:                                            # a class the camera cannot see -> false negative
            for extra in oe.CLASSES:
                if extra not in poly_true:
                    gt[extra] = [{'x': 10.0, 'y': 20.0}, {'x': 30.0, 'y': 25.0}]
                    break
        if seed == 7:
            gt = mirror_labels(gt)                                   # left/right swapped annotation: the mirrored pass wins
        poly_pred = get_polylines(js_pred, 960, 540, sampling_factor=0.9)
        c1, pc1, er1 = evaluate_camera_prediction(poly_pred, gt, 5)
        c2, pc2, er2 = evaluate_camera_prediction(poly_pred, mirror_labels(gt), 5)
        a1 = c1[0, 0] / c1.sum() if c1.sum() > 0 else 0.
        a2 = c2[0, 0] / c2.sum() if c2.sum() > 0 else 0.
        # the oracle on the same camera / annotations
        from oracle.camera_math import rotation_from_ptr
        R = rotation_from_ptr(np.deg2rad(js_pred['pan_degrees']), np.deg2rad(js_pred['tilt_degrees']), np.deg2rad(js_pred['roll_degrees']))
        pos = np.array(js_pred['position_meters'])
        mine = oe.get_polylines(pos, R, js_pred['x_focal_length'], js_pred['y_focal_length'], tuple(js_pred['principal_point']), 960, 540)
        assert list(mine) == list(poly_pred), (list(mine), list(poly_pred))
        for k in mine:
            a = np.array([[p['x'], p['y']] for p in poly_pred[k]])
            assert a.shape == np.array(mine[k]).shape and np.array_equal(a, np.array(mine[k])), k
        gt_t = {k: [(p['x'], p['y']) for p in v] for k, v in gt.items()}
        oc, oa, o1, o2 = oe.evaluate_frame(pos, R, js_pred['x_focal_length'], js_pred['y_focal_length'],
                                           tuple(js_pred['principal_point']), gt_t, 5)
        assert np.array_equal(o1, c1) and np.array_equal(o2, c2) and oa == max(a1, a2), (o1, c1, o2, c2)
        # per-class point confusions and reprojection errors (evaluate_camera.py:172-226), both label orientations
        for tag, labels, pc_ref, er_ref in (('1', gt_t, pc1, er1), ('2', oe.mirror_labels(gt_t), pc2, er2)):
            _, pc_o, er_o = oe.evaluate_camera_prediction(mine, labels, 5, detail=True)
            assert set(pc_o) == set(pc_ref) and set(er_o) == set(er_ref)
            for k in pc_ref:
                assert np.array_equal(pc_o[k], pc_ref[k]), k
                out[f'{n}.pc{tag}.{k}'] = np.asarray(pc_ref[k], dtype=np.float64)
            for k in er_ref:
                assert np.array_equal(np.array(er_o[k]), np.array(er_ref[k])), k
                out[f'{n}.err{tag}.{k}'] = np.array(er_ref[k], dtype=np.float64)
            out[f'{n}.pc{tag}.classes'] = np.array(sorted(pc_ref))
            out[f'{n}.err{tag}.classes'] = np.array(sorted(er_ref))
        out[f'{n}.position'] = pos; out[f'{n}.rotation'] = R
        out[f'{n}.f'] = np.array([js_pred['x_focal_length'], js_pred['y_focal_length']])
        out[f'{n}.pp'] = np.array(js_pred['principal_point'])
        out[f'{n}.gt_classes'] = np.array(list(gt_t))
        for k, v in gt_t.items():
            out[f'{n}.gt.{k}'] = np.array(v)
        out[f'{n}.conf1'] = c1; out[f'{n}.conf2'] = c2; out[f'{n}.acc'] = np.array([a1, a2])
        out[f'{n}.npoly'] = np.array([len(poly_pred.get(c, [])) for c in oe.CLASSES])
        n += 1
    out['n'] = np.array(n)
    np.savez_compressed(os.path.join(GOLD, 'evaluator_batch.npz'), **out)
    print('evaluator_batch ok:', n, 'frames; accuracies', [tuple(np.round(out[f"{i}.acc"], 3)) for i in range(n)])


def jpeg_test_image(h, w, seed):
    """A frame with what a broadcast frame has: smooth gradients (grass), sharp white lines, saturated patches
    (range limiting / colour clamping), noise."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    img = np.stack([40 + 30 * np.sin(xx / 37.0 + yy / 53.0), 130 + 60 * np.cos(xx / 29.0) * np.sin(yy / 41.0),
                    50 + (xx * 3 + yy * 5) % 97], -1).astype(np.float64)
    img += rng.normal(0, 5, img.shape)
    for k in range(4):
        img[np.abs(yy - (xx * (0.3 + 0.2 * k) + 5 * k)) < 1.2] = 255          # field lines
    img[h // 3:h // 2, w // 4:w // 2] = [255, 0, 0]
    img[h // 2:h // 2 + max(1, h // 6), w // 2:w // 2 + max(1, w // 5)] = [0, 0, 255]
    img[: max(1, h // 8), : max(1, w // 8)] = 0
    return np.clip(img, 0, 255).astype(np.uint8)


def gen_jpeg():
    """N3: JPEG byte streams (Pillow's encoder) and the pixels libjpeg-turbo decodes from them (Pillow's decoder,
    library defaults JDCT_ISLOW + fancy upsampling = what cv2.imread of make_submit.py:62 runs), stored BGR as cv2
    returns them.  cv2 itself is not installed in the build image."""
    import io
    from PIL import Image, features
    assert features.check_feature('libjpeg_turbo')
    out = {}
    names = []
    k = 0
    for (h, w) in [(48, 64), (29, 37), (16, 16), (8, 8), (7, 5), (1, 1), (33, 3), (17, 4), (100, 130)]:
        for sub in ('444', '422', '420', 'gray'):
            for q, rs in ((95, 0), (60, 3), (20, 0)):
                if q == 20 and (h, w) not in ((48, 64), (29, 37)):
                    continue
                im = Image.fromarray(jpeg_test_image(h, w, k))
                kw = dict(quality=q)
                if sub == 'gray':
                    im = im.convert('L')
                else:
                    kw['subsampling'] = {'444': 0, '422': 1, '420': 2}[sub]
                if rs:
                    kw['restart_marker_blocks'] = rs
                b = io.BytesIO()
                im.save(b, 'JPEG', **kw)
                data = b.getvalue()
                ref = np.asarray(Image.open(io.BytesIO(data)).convert('RGB'))[..., ::-1]
                name = f'{h}x{w}_{sub}_q{q}_r{rs}'
                names.append(name)
                out['jpg.' + name] = np.frombuffer(data, np.uint8)
                out['bgr.' + name] = np.ascontiguousarray(ref)
                k += 1
    # one frame at the reference's size (C3: 960x540), 4:2:0 as video-frame extraction writes it
    im = Image.fromarray(jpeg_test_image(540, 960, 1000))
    b = io.BytesIO()
    im.save(b, 'JPEG', quality=85, subsampling=2)
    data = b.getvalue()
    ref = np.asarray(Image.open(io.BytesIO(data)).convert('RGB'))[..., ::-1]
    out['jpg.full'] = np.frombuffer(data, np.uint8)
    out['bgr.full.rowsum'] = ref.astype(np.int64).sum(axis=(1, 2))
    out['bgr.full.colsum'] = ref.astype(np.int64).sum(axis=(0, 2))
    out['bgr.full.tile'] = np.ascontiguousarray(ref[256:304, 448:512])
    # a progressive file: must be refused
    b = io.BytesIO()
    Image.fromarray(jpeg_test_image(32, 32, 5)).save(b, 'JPEG', quality=80, progressive=True)
    out['jpg.progressive'] = np.frombuffer(b.getvalue(), np.uint8)
    out['names'] = np.array(names)
    np.savez_compressed(os.path.join(GOLD, 'jpeg_cases.npz'), **out)
    print('jpeg ok:', len(names), 'cases,', os.path.getsize(os.path.join(GOLD, 'jpeg_cases.npz')) // 1024, 'KiB; full-size',
          len(data), 'bytes')


def gen_target():
    """N4: HRNetLoss.create_target of the imported reference (loss.py:81-87) on keypoints that cover its visibility quirk
    (any component == 1), points outside the canvas, sub-pixel centres, and the training configuration
    (sigma 3, stride 2, pred_size 270x480; train_config.yaml:38-40) -- stored sparsely (a few planes + checksums)."""
    import torch
    from src.models.hrnet.loss import HRNetLoss
    from oracle import synth
    rng = np.random.default_rng(77)
    out = {}
    cases = {'small': (3, 7, 2.0, (20, 33)), 'train': (2, 57, 3.0, (270, 480))}
    for name, (B, N, sigma, hw) in cases.items():
        kp = np.zeros((B, N, 3), np.float32)
        kp[..., 0] = rng.uniform(-5, hw[1] + 5, (B, N))
        kp[..., 1] = rng.uniform(-5, hw[0] + 5, (B, N))
        kp[..., 2] = (rng.uniform(size=(B, N)) < 0.7).astype(np.float32)
        kp[0, 0] = (1.0, 7.25, 0.0)          # x == 1 makes it "visible" although the flag is 0 (loss.py:49)
        kp[0, 1] = (5.5, 1.0, 0.0)           # y == 1 likewise
        kp[0, 2] = (4.0, 6.0, 0.0)           # really invisible
        kp[0, 3] = (10.0, 3.0, 1.0)
        loss = HRNetLoss(num_refinement_stages=0, sigma=sigma, stride=1, pred_size=hw, num_keypoints=N)
        ref = loss.create_target(torch.from_numpy(kp)).numpy()
        mine = synth.create_target(kp, sigma, hw)
        ulp = np.abs(ref.view(np.int32).astype(np.int64) - mine.view(np.int32).astype(np.int64))
        normal = np.abs(ref) > 1e-30          # torch's vectorised exp flushes / rounds differently in the denormal range
        normal[:, N] = False                  # background = 1 - max: an ulp of the maximum is many ulps of the difference
        assert ulp[normal].max() <= 4 and np.abs(ref - mine)[~normal].max() <= 2.4e-7, (ulp[normal].max(), np.abs(ref - mine)[~normal].max())
        ulp = ulp[normal]
        out[f'{name}.kp'] = kp
        out[f'{name}.sigma'] = np.float32(sigma)
        out[f'{name}.hw'] = np.array(hw)
        if name == 'small':
            out[f'{name}.target'] = ref
        else:
            out[f'{name}.planes'] = ref[:, [0, 1, 2, 3, 30, N]]
            out[f'{name}.chan_sum'] = ref.astype(np.float64).sum(axis=(2, 3))
        print('target', name, 'ok: max ulp distance oracle vs reference', int(ulp.max()))
    np.savez_compressed(os.path.join(GOLD, 'target.npz'), **out)


def gen_annotations():
    """N4 geometry: the reference's get_intersections (src/datatools/intersections.py:99-124 -> ellipse.py:326-400) on synthetic
    annotations.  Its two third-party calls are replaced by the build's own routines -- `ellipse.LsqEllipse().fit(X).coefficients`
    by annotations.fit_ellipse, `cv2.findHomography(src, dst, RANSAC, thr)` by annotations.homography_ransac -- so the capture
    pins everything the reference computes in numpy: line fits and the recursive intersection, tangent points, circle x line
    with the local refinement, the Top / Bottom choice, the image-bounds filter and the control flow around them."""
    import sncal_amd
    from sncal_amd import annotations as an

    class Fit:
        def fit(self, X):
            q = an.fit_ellipse(np.asarray(X, dtype=np.float64))
            self.coefficients = list(q) if q is not None else []
            return self
    sys.modules['ellipse'].LsqEllipse = Fit
    sys.modules['cv2'].findHomography = lambda src, dst, method, thr: (an.homography_ransac(src, dst, thr), None)
    import src.datatools.ellipse as rel
    rel.LsqEllipse = Fit
    from src.datatools.intersections import get_intersections
    cases = []
    for seed in range(24):
        pts, _ = sncal_amd.synth.synthetic_annotation(seed)
        if seed % 5 == 4:                                   # drop a few classes: exercises the homography fill / the mask
            for k in list(pts)[::3]:
                pts.pop(k)
        with contextlib.redirect_stdout(io.StringIO()):
            ref, mask = get_intersections({k: list(v) for k, v in pts.items()})
        mine, mmask = an.get_intersections(pts)
        worst = 0.0
        for i in range(57):
            r, m = ref.get(i), mine.get(i)
            assert (r is None) == (m is None), (seed, i, r, m)
            if r is not None:
                worst = max(worst, abs(r[0] - m[0]), abs(r[1] - m[1]))
        assert sorted(mask) == sorted(mmask), (seed, mask, mmask)
        assert worst < 1e-6, (seed, worst)
        arr = np.full((57, 2), np.nan)
        for i in range(57):
            if ref.get(i) is not None:
                arr[i] = ref[i]
        cases.append({'points': {k: [list(map(float, p)) for p in v] for k, v in pts.items()}, 'labels': arr.tolist(), 'mask': sorted(map(int, mask))})
    with open(os.path.join(GOLD, 'annotations.json'), 'w') as f:
        json.dump(cases, f)
    n = sum(int(np.isfinite(np.array(c['labels'])[:, 0]).sum()) for c in cases)
    print('annotations ok:', len(cases), 'frames,', n, 'labels')


def gen_hrnet(name, cfg_name, hw, seed, head_gain, line=False, store_full=True, batch=1):
    from oracle import hrnet_ref as hr, decode as od
    cfg = hr.load_config(cfg_name)
    if line:
        from src.models.line.model import HRNetHeatmap
    else:
        from src.models.hrnet.model import HRNetHeatmap
    torch.manual_seed(0)
    net = HRNetHeatmap(ref_cfg(cfg), num_refinement_stages=0, num_heatmaps=cfg['num_classes'])
    sd = hr.seeded_state_dict(cfg, seed, head_gain)
    missing = net.load_state_dict(sd, strict=True)       # validates names + shapes of the enumerator
    net.eval()
    x = hr.seeded_input(batch, hw[0], hw[1], seed + 1)
    with torch.no_grad():
        ref = net(x)[-1]
    mine, inter = hr.forward(sd, x, cfg, return_intermediates=True)
    err = (ref - mine).abs().max().item()
    assert err < 1e-4, err
    refn = ref.numpy()
    out = {'hw': np.array(hw), 'seed': np.array(seed), 'head_gain': np.array(head_gain), 'batch': np.array(batch),
           'checksum': np.array(checksum(refn)), 'abs_checksum': np.array(checksum(np.abs(refn))),
           'n_params': np.array(sum(int(v.numel()) for k, v in sd.items() if 'num_batches' not in k)),
           'macs': np.array(hr.conv_macs(cfg, hw[0], hw[1]))}
    if line:
        tr_out = od.line_decode(refn, 3.0, 4.0)
        from src.models.line.transforms import EHMPredictionTransform
        ref_dec = EHMPredictionTransform(scale=4, sigma=3)(ref.clone()).numpy()
        assert np.array_equal(ref_dec[..., :2], tr_out[..., :2])
        out['decode'] = ref_dec
        out['maxp'] = refn.max(axis=(2, 3))
    else:
        from src.models.hrnet.transforms import HRNetPredictionTransform
        ref_dec = HRNetPredictionTransform((540, 960))(ref).numpy()
        assert np.array_equal(ref_dec[..., :2], od.keypoint_decode(refn, (540, 960))[..., :2])
        out['decode'] = ref_dec
        p = np.exp(refn)
        flat = np.sort(p.reshape(p.shape[0], p.shape[1], -1), axis=-1)
        out['maxp'] = flat[..., -1]
        out['gap'] = (flat[..., -1] - flat[..., -2])          # top-1 / top-2 separation per channel
    if store_full:
        out['out'] = refn
    else:
        out['out_strided'] = refn[:, :, ::16, ::16].copy()
    for k in ('stage2', 'stage3', 'stage4'):
        for bi, t in enumerate(inter[k]):
            out[f'{k}.{bi}.checksum'] = np.array(checksum(t.numpy()))
            out[f'{k}.{bi}.abs'] = np.array(checksum(np.abs(t.numpy())))
            if store_full:
                out[f'{k}.{bi}'] = t.numpy()[:, ::4].copy()      # every 4th channel keeps it small
    np.savez_compressed(os.path.join(GOLD, name + '.npz'), **out)
    print(name, 'ok: max|ref-oracle| =', err, 'params', int(out['n_params']), 'GMAC', int(out['macs']) / 1e9,
          'maxp range', float(out['maxp'].min()), float(out['maxp'].max()),
          ('min gap %.3g' % float(out['gap'].min())) if 'gap' in out else '')


if __name__ == '__main__':
    os.makedirs(GOLD, exist_ok=True)
    install_stubs()
    which = sys.argv[1:] or ['pitch', 'decode', 'camera', 'annotations', 'lines', 'evaluator', 'evaluator_batch', 'jpeg', 'target', 'hrnet']
    if 'pitch' in which:
        gen_pitch()
    if 'decode' in which:
        gen_decode()
    if 'camera' in which:
        gen_camera()
    if 'annotations' in which:
        gen_annotations()
    if 'lines' in which:
        gen_lines()
    if 'evaluator_batch' in which:
        gen_evaluator_batch()
    if 'evaluator' in which:
        gen_evaluator()
    if 'jpeg' in which:
        gen_jpeg()
    if 'target' in which:
        gen_target()
    if 'hrnet' in which:
        gen_hrnet('hrnet_w18_64x96', 'hrnet_w18', (64, 96), 3, 4.0)
        gen_hrnet('hrnet_w18_135x240', 'hrnet_w18', (135, 240), 5, 4.0, store_full=False, batch=2)
        gen_hrnet('line_w18_64x96', 'line_hrnet_w18', (64, 96), 4, 4.0, line=True)
        gen_hrnet('hrnet_w48_540x960', 'hrnet_w48', (540, 960), 1, 1.5, store_full=False)
        gen_hrnet('line_w48_540x960', 'line_hrnet_w48', (540, 960), 2, 1.5, line=True, store_full=False)
