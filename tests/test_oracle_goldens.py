"""CPU: the oracle/ restatements against the golden vectors captured from the imported reference
(tools/make_golden.py).  This is what pins the oracle."""
import json
import os

import numpy as np
import pytest

from oracle import camera_math as cm
from oracle import decode as od
from oracle import lines as ol
from oracle import pitch as op
from oracle import synth


GAUSS_CASES = {'gauss_68x120': ((68, 120), 1e-12), 'gauss_135x240_neginf': ((135, 240), 0)}


def regen_gauss_case(g, case):
    hw, floor = GAUSS_CASES[case]
    return synth.synth_logp(list(g['seeds.' + case]), hw=hw, floor=floor)[0]


def test_pitch_template(gold_dir):
    g = np.load(os.path.join(gold_dir, 'pitch.npz'))
    assert np.abs(g['points'] - op.pitch_points()).max() < 1e-12
    assert list(g['keep_points']) == op.KEEP_POINTS and 29 not in op.KEEP_POINTS   # Q7
    assert list(g['goal_left']) == op.GOAL_LEFT and list(g['goal_right']) == op.GOAL_RIGHT
    assert list(g['top_gates']) == op.TOP_GATES
    assert list(g['line_cls']) == ol.LINE_CLS
    assert [tuple(p) for p in g['line_pairs']] == [ol.LINE_INTERSECTIONS[i] for i in range(30)]


@pytest.mark.parametrize('case', ['gauss_68x120', 'gauss_135x240_neginf', 'ties_34x60', 'noise_34x60'])
def test_keypoint_decode_golden(gold_dir, case):
    g = np.load(os.path.join(gold_dir, 'decode_keypoints.npz'))
    lp = g[case + '.in']
    if lp.size == 0:   # large inputs are regenerated from their seeds
        lp = regen_gauss_case(g, case)
    out = od.keypoint_decode(lp, (540, 960))
    ref = g[case + '.out']
    assert np.array_equal(out[..., :2], ref[..., :2])            # indices: bit-identical
    assert np.allclose(out[..., 2], ref[..., 2], rtol=2e-7, atol=0)   # conf: 1 ULP (exp implementation)


def test_tie_rule_is_separable(gold_dir):
    """H1: x and y come from separate reductions and need not belong to one pixel."""
    g = np.load(os.path.join(gold_dir, 'decode_keypoints.npz'))
    out = g['ties_34x60.out'][0]
    assert tuple(out[0, :2]) == (7 * 960 / 60, 10 * 540 / 34)     # col of one maximum, row of the other
    assert tuple(out[1, :2]) == (0.0, 0.0)                         # all-equal channel
    assert tuple(out[2, :2]) == (5 * 16.0, np.float32(2 * 540) / np.float32(34))   # exp() collision at 1.0


@pytest.mark.parametrize('sigma', [3, 6])
def test_line_decode_golden(gold_dir, sigma):
    g = np.load(os.path.join(gold_dir, 'decode_lines.npz'))
    out = od.line_decode(g['heat'], float(sigma), 4.0)
    ref = g[f'out_sigma{sigma}']
    assert np.array_equal(out[..., :2], ref[..., :2])
    assert np.allclose(out[..., 2], ref[..., 2], rtol=1e-5, atol=1e-7)


def test_camera_math_golden(gold_dir):
    g = np.load(os.path.join(gold_dir, 'camera.npz'), allow_pickle=False)
    P = op.pitch_points()
    for i in range(int(g['n'])):
        pos, R, f = g[f'{i}.position'], g[f'{i}.rotation'], float(g[f'{i}.f'])
        proj = np.stack([cm.project_point(pos, R, f, f, (480., 270.), p) for p in P])
        assert np.abs(proj - g[f'{i}.proj']).max() < 1e-9
        js = json.loads(str(g[f'{i}.json']))
        mine = cm.to_json(pos, R, f, f, (480., 270.))
        for k in ('pan_degrees', 'tilt_degrees', 'roll_degrees'):
            assert abs(js[k] - mine[k]) < 1e-10
        Rj = cm.rotation_from_ptr(*np.deg2rad([js['pan_degrees'], js['tilt_degrees'], js['roll_degrees']]))
        assert np.abs(Rj - g[f'{i}.rot_from_json']).max() < 1e-12
        ids = g[f'{i}.obs_ids']
        assert abs(cm.projection_rmse(pos, R, f, f, (480., 270.), P[ids], g[f'{i}.obs']) - float(g[f'{i}.rmse'])) < 1e-9
        ok, fx, fy = cm.k_from_plane_homography(g[f'{i}.H'])
        assert ok == bool(g[f'{i}.k_ok'])
        if ok:
            assert abs(fx - float(g[f'{i}.k_fx'])) < 1e-6 * fx and abs(fy - float(g[f'{i}.k_fy'])) < 1e-6 * fy
            assert abs(fx - f) < 1e-3 * f      # exact homography -> the true focal length comes back
        assert cm.good_camera(f, pos) == bool(g[f'{i}.good'])


def test_line_join_golden(gold_dir):
    with open(os.path.join(gold_dir, 'lines.json')) as f:
        g = json.load(f)
    pts = ol.lines_to_keypoints({k: tuple(v) for k, v in g['lines'].items()})
    assert {str(k) for k in pts} == set(g['keypoints'])
    for k, v in pts.items():
        assert np.allclose(v, g['keypoints'][str(k)])
    assert ol.line_eq_intersection((1.0, 0.0), (1.00001, 5.0)) is None
    assert ol.slope_intercept((1., 2.), (1., 2.)) == (None, None)
    # float32 peak coordinates: differences in float32, float64 from `+ delta` on (numpy 1.24.2, the reference's pin);
    # the golden lines were captured from the reference function under exactly that arithmetic
    gd = np.load(os.path.join(gold_dir, 'decode_lines.npz'))
    hl = gd['out_sigma3'][:1] / np.array([4, 4, 1], dtype=np.float32)
    lines, _ = ol.get_line_data(hl, scale=4, prob_thre=0.2)
    ref = {k: v for k, v in g['lines'].items() if k != 'Goal left post left'}
    for k, v in lines.items():
        if k in ref:
            assert (float(v[0]), float(v[1])) == (ref[k][0], ref[k][1])
    arr = ol.keypoints_array(hl)
    assert arr.shape == (1, 30, 3) and arr[0, :, 2].sum() >= 10


def _eval_frames(gold_dir):
    g = np.load(os.path.join(gold_dir, 'evaluator_batch.npz'))
    frames = []
    for i in range(int(g['n'])):
        gt = {str(c): [tuple(p) for p in g[f'{i}.gt.{c}']] for c in g[f'{i}.gt_classes']}
        detail = {}
        for tag in ('1', '2'):
            detail['pc' + tag] = {str(c): g[f'{i}.pc{tag}.{c}'] for c in g[f'{i}.pc{tag}.classes']}
            detail['err' + tag] = {str(c): g[f'{i}.err{tag}.{c}'] for c in g[f'{i}.err{tag}.classes']}
        frames.append(dict(position=g[f'{i}.position'], rotation=g[f'{i}.rotation'], f=g[f'{i}.f'], pp=g[f'{i}.pp'], gt=gt,
                           conf1=g[f'{i}.conf1'], conf2=g[f'{i}.conf2'], acc=g[f'{i}.acc'], npoly=g[f'{i}.npoly'], **detail))
    return g, frames


def test_evaluator_oracle_matches_reference_capture(gold_dir):
    """N2 oracle: sampled pitch model, polylines, plain + mirrored confusion and accuracies of
    baseline/evaluate_camera.py, captured from the imported reference (tools/make_golden.py evaluator_batch)."""
    from oracle import evaluate as oe
    g, frames = _eval_frames(gold_dir)
    pts, start = oe.field_table()
    assert np.array_equal(pts, g['field_points']) and np.array_equal(start, g['class_start']) and list(g['classes']) == oe.CLASSES
    for fr in frames:
        poly = oe.get_polylines(fr['position'], fr['rotation'], fr['f'][0], fr['f'][1], tuple(fr['pp']), 960, 540, (pts, start))
        assert [len(poly.get(c, [])) for c in oe.CLASSES] == list(fr['npoly'])
        conf, acc, c1, c2 = oe.evaluate_frame(fr['position'], fr['rotation'], fr['f'][0], fr['f'][1], tuple(fr['pp']), fr['gt'], 5,
                                              table=(pts, start))
        assert np.array_equal(c1, fr['conf1']) and np.array_equal(c2, fr['conf2']) and acc == max(fr['acc'])
        for tag, labels in (('1', fr['gt']), ('2', oe.mirror_labels(fr['gt']))):      # per-class confusions and errors
            _, pc, er = oe.evaluate_camera_prediction(poly, labels, 5, detail=True)
            assert set(pc) == set(fr['pc' + tag]) and set(er) == set(fr['err' + tag])
            assert all(np.array_equal(pc[k], fr['pc' + tag][k]) for k in pc)
            assert all(np.array_equal(np.array(er[k]), fr['err' + tag][k]) for k in er)
    assert any(fr['acc'][1] > fr['acc'][0] for fr in frames)          # the mirrored pass wins somewhere


def _jpeg_cases(gold_dir):
    g = np.load(os.path.join(gold_dir, 'jpeg_cases.npz'))
    return g, [str(n) for n in g['names']]


def test_jpeg_oracle_matches_libjpeg_turbo_capture(gold_dir):
    """N3: the decode restatement (Huffman, islow IDCT, fancy upsampling, fixed-point colour) reproduces, byte for
    byte, what libjpeg-turbo decoded from the same streams -- every sampling layout, odd sizes down to 1x1, restart
    intervals, three qualities; plus the 960x540 frame through row/column sums and a tile."""
    from oracle import jpeg as oj
    g, names = _jpeg_cases(gold_dir)
    assert len(names) >= 70
    for n in names:
        got = oj.decode_bgr(g['jpg.' + n].tobytes())
        assert got.dtype == np.uint8 and np.array_equal(got, g['bgr.' + n]), n
    full = oj.decode_bgr(g['jpg.full'].tobytes())
    assert full.shape == (540, 960, 3)
    assert np.array_equal(full.astype(np.int64).sum(axis=(1, 2)), g['bgr.full.rowsum'])
    assert np.array_equal(full.astype(np.int64).sum(axis=(0, 2)), g['bgr.full.colsum'])
    assert np.array_equal(full[256:304, 448:512], g['bgr.full.tile'])
    with pytest.raises(oj.JpegError):
        oj.decode_bgr(g['jpg.progressive'].tobytes())


def _target_close(ref, got, n_kp):
    """Tolerance of the N4 target: keypoint channels within 4 ulp of the reference's torch values wherever those are
    normal numbers (torch's vectorised exp vs the correctly rounded exp, times two factors), background channel
    (1 - max) within 2 ulp of 1.0."""
    ulp = np.abs(ref.view(np.int32).astype(np.int64) - got.view(np.int32).astype(np.int64))
    normal = np.abs(ref) > 1e-30
    normal[:, n_kp] = False
    return ulp[normal].max() <= 4 and np.abs(ref - got)[~normal].max() <= 2.4e-7


def test_target_oracle_matches_reference_capture(gold_dir):
    """N4: HRNetLoss.create_target captured from the imported reference (tools/make_golden.py gen_target), including the
    `any(keypoints == 1)` visibility quirk of loss.py:49."""
    g = np.load(os.path.join(gold_dir, 'target.npz'))
    kp, sigma, hw = g['small.kp'], float(g['small.sigma']), tuple(g['small.hw'])
    got = synth.create_target(kp, sigma, hw)
    assert got.shape == g['small.target'].shape and _target_close(g['small.target'], got, kp.shape[1])
    assert got[0, 0].max() > 0.5 and got[0, 1].max() > 0.5 and not got[0, 2].any()      # x == 1 / y == 1 visible, flag 0 not
    kp, sigma, hw = g['train.kp'], float(g['train.sigma']), tuple(g['train.hw'])
    got = synth.create_target(kp, sigma, hw)
    n = kp.shape[1]
    planes = got[:, [0, 1, 2, 3, 30, n]]
    ref = g['train.planes']
    ulp = np.abs(ref.view(np.int32).astype(np.int64) - planes.view(np.int32).astype(np.int64))
    normal = np.abs(ref) > 1e-30
    normal[:, -1] = False
    assert ulp[normal].max() <= 4 and np.abs(ref - planes)[~normal].max() <= 2.4e-7
    assert np.allclose(got.astype(np.float64).sum(axis=(2, 3)), g['train.chan_sum'], rtol=1e-6, atol=1e-4)
