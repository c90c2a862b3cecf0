"""GPU: sncal_evaluate_cameras (batched accuracy@t metric) vs the reference capture and the oracle."""
import ctypes
import os

import numpy as np
import pytest
import torch

from oracle import evaluate as oe
from oracle import synth
from test_oracle_goldens import _eval_frames

pytestmark = pytest.mark.gpu


def _records(sncal, frames, cuda):
    """sncal_camera records for cameras given as (position, rotation, fx, fy) dicts; status 0 where None."""
    Cam = sncal._lib.Camera
    buf = (Cam * len(frames))()
    for i, fr in enumerate(frames):
        if fr is None:
            continue
        for k in range(3):
            buf[i].position[k] = float(fr['position'][k])
        for k in range(9):
            buf[i].rotation[k] = float(np.asarray(fr['rotation']).reshape(-1)[k])
        buf[i].fx, buf[i].fy = float(fr['f'][0]), float(fr['f'][1])
        buf[i].cx, buf[i].cy = 479.5, 269.5            # the solver's centre; the metric uses (w/2, h/2) like the JSON
        buf[i].status = 1
    raw = np.frombuffer(bytes(buf), dtype=np.uint8).reshape(len(frames), ctypes.sizeof(Cam)).copy()
    return torch.from_numpy(raw).to(cuda)


def test_evaluator_matches_reference_capture(sncal, cuda, gold_dir):
    g, frames = _eval_frames(gold_dir)
    ev = sncal.CameraEvaluator(cuda, 960, 540, threshold=5)
    frames_m = frames[:4] + [None] + frames[4:]                     # one missed frame (no camera)
    ann = [fr['gt'] if fr is not None else {} for fr in frames_m]
    ann[0] = dict(ann[0]); ann[0]['Goal unknown'] = [(5.0, 5.0)]       # a class the pitch model does not have: +1 FN
    out = ev.evaluate(_records(sncal, frames_m, cuda), ann).cpu().numpy()
    k = 0
    for i, fr in enumerate(frames_m):
        if fr is None:
            assert out[i, 11] == 0 and not out[i].any()
            continue
        c1, c2 = fr['conf1'].copy(), fr['conf2'].copy()
        if i == 0:
            c1[1, 0] += 1; c2[1, 0] += 1
        assert np.array_equal(out[i, 0:4], c1.reshape(-1)) and np.array_equal(out[i, 4:8], c2.reshape(-1)), (i, out[i], c1, c2)
        a1 = c1[0, 0] / c1.sum() if c1.sum() > 0 else 0.
        a2 = c2[0, 0] / c2.sum() if c2.sum() > 0 else 0.
        assert out[i, 8] == np.float32(a1) and out[i, 9] == np.float32(a2) and out[i, 10] == (1 if a1 > a2 else 2) and out[i, 11] == 1
        k += 1
    s = sncal.CameraEvaluator.summarize(out)
    assert abs(s['completeness'] - 10 / 11) < 1e-12 and 0 < s['accuracy'] < 1 and abs(s['final_score'] - s['completeness'] * s['accuracy']) < 1e-12


def test_per_class_confusions_and_errors_match_reference_capture(sncal, cuda, gold_dir):
    """sncal_evaluate_cameras_detail: `per_class_confusion` exactly and `dict_errors` (fp64 distances) to 1e-9 relative
    against what evaluate_camera_prediction returned in the reference, both label orientations; class_report
    accumulates them as evaluate_camera.py:335-373 does."""
    g, frames = _eval_frames(gold_dir)
    ev = sncal.CameraEvaluator(cuda, 960, 540, threshold=5)
    frames_m = frames[:3] + [None] + frames[3:]
    ann = [fr['gt'] if fr is not None else {} for fr in frames_m]
    out, err, cls = ev.evaluate(_records(sncal, frames_m, cuda), ann, detail=True)
    assert torch.isnan(err[3]).all() and not cls[3].any()
    plain = ev.evaluate(_records(sncal, frames_m, cuda), ann)
    assert torch.equal(plain, out)
    n_err = 0
    for i, fr in enumerate(frames_m):
        if fr is None:
            continue
        for which in (1, 2):
            pc, er = ev.frame_detail(out, err, cls, ann, i, which)
            want_pc, want_er = fr[f'pc{which}'], fr[f'err{which}']
            assert set(pc) == set(want_pc) and set(er) == set(want_er), (i, which)
            for k in pc:
                assert np.array_equal(pc[k], want_pc[k]), (i, which, k)
            for k in er:
                assert np.allclose(er[k], want_er[k], rtol=1e-9, atol=1e-9), (i, which, k)
                n_err += len(er[k])
    assert n_err > 200
    report, errors, hist = ev.class_report(out, err, cls, ann)
    acc_conf = {}
    for i, fr in enumerate(frames_m):
        if fr is None:
            continue
        which = 1 if fr['acc'][0] > fr['acc'][1] else 2
        for k, m in fr[f'pc{which}'].items():
            acc_conf[k] = acc_conf.get(k, 0) + m
    assert set(report) == set(acc_conf)
    for k in report:
        assert np.array_equal(report[k]['confusion'], acc_conf[k])
        assert sum(hist[k][0]) <= len(errors[k]) if k in hist else True


def test_evaluator_matches_oracle_on_random_cameras(sncal, cuda):
    """64 synthetic cameras (synth.sample_camera) with perturbed predictions and subsampled annotations."""
    rng = np.random.default_rng(3)
    table = oe.field_table()
    frames, ann, expect = [], [], []
    for s in range(64):
        cam = synth.sample_camera(np.random.Generator(np.random.PCG64(5000 + s)))
        true_poly = oe.get_polylines(cam['position'], cam['rotation'], cam['f'], cam['f'], (480., 270.), 960, 540, table)
        pred = dict(position=cam['position'] + rng.normal(0, 0.15, 3), rotation=cam['rotation'], f=(cam['f'] * (1 + rng.normal(0, 0.004)),) * 2)
        gt = {c: [(x + rng.normal(0, 1.0), y + rng.normal(0, 1.0)) for (x, y) in v[::5][:8]] for c, v in true_poly.items() if rng.uniform() > 0.2}
        frames.append(pred); ann.append(gt)
        expect.append(oe.evaluate_frame(pred['position'], pred['rotation'], pred['f'][0], pred['f'][1], (480., 270.), gt, 5, table=table))
    ev = sncal.CameraEvaluator(cuda)
    out = ev.evaluate(_records(sncal, frames, cuda), ann).cpu().numpy()
    bad = 0
    for i, (conf, acc, c1, c2) in enumerate(expect):
        if not (np.array_equal(out[i, 0:4], c1.reshape(-1)) and np.array_equal(out[i, 4:8], c2.reshape(-1))):
            bad += 1
    assert bad == 0, f'{bad} of 64 frames differ from the oracle'
    accs = np.where(out[:, 10] == 1, out[:, 8], out[:, 9])
    assert np.allclose(accs, [a for _, a, _, _ in expect], rtol=1e-6) and 0.1 < accs.mean() < 0.99


def test_solved_cameras_score_on_the_benchmark_metric(sncal, cuda):
    """The chain the reference runs across three scripts (predict -> CameraCreator -> evaluate_camera.py), on the device
    end to end: 48 synthetic cameras -> noisy template keypoints -> sncal_calibrate -> sncal_evaluate_cameras against
    annotations sampled from the TRUE cameras' pitch polylines.  With 1 px keypoint noise the solved cameras must
    reproduce the true polylines within the benchmark's 5 px on nearly every class (mean accuracy@5 > 0.9), and the
    oracle's metric of the same solved cameras must agree with the kernel's."""
    rng = np.random.Generator(np.random.PCG64(2024))
    table = oe.field_table()
    kps, ann = [], []
    while len(kps) < 48:
        cam = sncal.synth.random_camera(rng)
        kp = sncal.synth.keypoints_for_camera(cam, rng, sigma_px=1.0, outlier_frac=0.0)
        if (kp[:, 2] > 0.5).sum() < 14:
            continue
        true_poly = oe.get_polylines(np.asarray(cam.position), np.asarray(cam.rotation), cam.xfocal_length, cam.yfocal_length,
                                     (480., 270.), 960, 540, table)
        if len(true_poly) < 4:
            continue
        kps.append(kp)
        ann.append({c: [tuple(p) for p in v[::max(1, len(v) // 6)][:8]] for c, v in true_poly.items()})
    cc = sncal.submit.default_calibrator()
    rec = cc.solve_device(torch.from_numpy(np.stack(kps)).to(cuda))
    ev = sncal.CameraEvaluator(cuda, 960, 540, threshold=5)
    out = ev.evaluate(rec, ann).cpu().numpy()
    s = sncal.CameraEvaluator.summarize(out)
    assert s['completeness'] > 0.95 and s['accuracy'] > 0.9, s
    # the oracle's metric on the same solved cameras
    cams = [sncal.prediction.camera_from_record(r, cc.img_size) for r in cc.records(rec)]
    agree = 0
    for i, c in enumerate(cams):
        if c is None:
            assert out[i, 11] == 0
            continue
        conf, acc, c1, c2 = oe.evaluate_frame(np.asarray(c.position), np.asarray(c.rotation), c.xfocal_length, c.yfocal_length,
                                              (480., 270.), ann[i], 5, table=table)
        agree += int(np.array_equal(out[i, 0:4], c1.reshape(-1)) and np.array_equal(out[i, 4:8], c2.reshape(-1)))
    assert agree >= len([c for c in cams if c is not None]) - 1        # a distance within rounding of the threshold may flip one class
