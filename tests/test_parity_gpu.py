"""GPU: parity of the BENCHMARKED (bf16) engine against the exact-fp32 engine at the metric's own size (HRNet-W48,
960x540), on PEAKED heatmaps (sncal_amd.synth.peaked_state_dict / stamped_frames: the random network plus one
matched-filter signal path, see synth.py), with the decoded keypoints driving the camera solve.

north_star: "bit-identical keypoint indices on the same frames", "within 1e-4 relative on reprojection error".
The fp32 engine is the one pinned to the reference capture (tests/test_hrnet_gpu.py); this file measures how far the
bf16 engine is from it and writes the table the bench line's `parity` object summarises.
Solve parity is against the build's own oracle only -- OpenCV parity unpinned.
"""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

GAP_EDGES = [0.0, 0.02, 0.05, 0.1, 0.2, 0.5, 1.0, 2.0, 5.0, np.inf]       # top-1 / top-2 gap buckets, in log-probability
GAP_CERTAIN = 0.1        # stated gap: above it the bf16 engine must reproduce the fp32 index in 100 % of the cases
                         # (measured round 2: 100 % from 0.05 up at every sharpness, flips only below 0.02-0.05)
KW = dict(conf_thresh=0.5, conf_threshs=[0.5, 0.35, 0.2], algorithm='iterative_voter', max_rmse=55.0, max_rmse_rel=5.0,
          min_points=5, min_focal_length=10.0, min_points_per_plane=6, min_points_for_refinement=6, reliable_thresh=57)


def _weights(sncal):
    import bench
    return bench.seeded_weights('hrnet_w48', seed=1)


def _axis_gaps(logp):
    """logp (B,58,h,w) -> per (b, c<57): gap between the best and second-best column maximum / row maximum."""
    col = logp.max(axis=2)[:, :57]            # (B,57,w) max over rows
    row = logp.max(axis=3)[:, :57]            # (B,57,h)
    out = []
    for m in (col, row):
        s = np.sort(m, axis=-1)
        out.append(s[..., -1] - s[..., -2])
    return out                                 # [gap_x (B,57), gap_y (B,57)]


def _run(sncal, cuda, sd, x, dtype, want_heat):
    net = sncal.HRNetHeatmap('hrnet_w48', dtype=dtype, device=cuda)
    net.load_state_dict(sd)
    heat, kp = net.forward(x, want_heat=want_heat, decode_size=(540, 960))
    return (heat.cpu().numpy() if heat is not None else None), kp.cpu().numpy()


@pytest.mark.parametrize('peak_logit,noise_gain,must_agree', [(12.0, 0.25, True), (8.0, 0.5, False), (5.0, 1.0, False)])
def test_bf16_engine_index_agreement_on_peaked_heatmaps(sncal, cuda, peak_logit, noise_gain, must_agree):
    B = 8
    sd = sncal.synth.peaked_state_dict(_weights(sncal), peak_logit=peak_logit, noise_gain=noise_gain)
    frames, expect = sncal.synth.stamped_frames(B, seed=4242)
    x = torch.from_numpy(frames).to(cuda)
    heat32, kp32 = _run(sncal, cuda, sd, x, 'fp32', True)
    _, kp16 = _run(sncal, cuda, sd, x, 'bf16', False)
    gx, gy = _axis_gaps(heat32)
    same_x, same_y = kp32[..., 0] == kp16[..., 0], kp32[..., 1] == kp16[..., 1]
    table = []
    for lo, hi in zip(GAP_EDGES[:-1], GAP_EDGES[1:]):
        sel = np.concatenate([((gx >= lo) & (gx < hi)).ravel(), ((gy >= lo) & (gy < hi)).ravel()])
        same = np.concatenate([same_x.ravel(), same_y.ravel()])
        table.append({'gap_lo': lo, 'gap_hi': None if np.isinf(hi) else hi, 'n': int(sel.sum()),
                      'agreement': None if not sel.any() else round(float(same[sel].mean()), 6)})
    vis = expect[..., 2] > 0
    usable = kp32[..., 2] >= 0.2
    same = same_x & same_y
    summary = {'peak_logit': peak_logit, 'noise_gain': noise_gain, 'frames': B,
               'fp32_on_stamped_cell': round(float((kp32[..., :2] == expect[..., :2]).all(-1)[vis].mean()), 6),
               'visible': int(vis.sum()), 'usable_fp32_conf_ge_0.2': int(usable.sum()),
               'index_agreement_usable': round(float(same[usable].mean()), 6) if usable.any() else None,
               'index_agreement_all_rows': round(float(same.mean()), 6),
               'conf_delta_max_usable': round(float(np.abs(kp32[..., 2] - kp16[..., 2])[usable].max()), 6) if usable.any() else None,
               'buckets': table}
    # the solve on both engines' keypoints
    cc = sncal.CameraCreator(sncal.PITCH_POINTS, **KW)
    c32, c16 = cc.solve_batch(kp32), cc.solve_batch(kp16)
    deltas, frames_equal = [], 0
    for b in range(B):
        sel = (kp32[b, :, 2] > 0.2) | (kp16[b, :, 2] > 0.2)
        if same[b][sel].all():
            frames_equal += 1
            assert (c32[b] is None) == (c16[b] is None), b
            if c32[b] is not None:
                deltas.append(abs(c16[b].rmse - c32[b].rmse) / c32[b].rmse)
    summary.update(frames_with_identical_usable_indices=frames_equal, cameras_fp32=sum(c is not None for c in c32),
                   cameras_bf16=sum(c is not None for c in c16), rmse_rel_delta_max=max(deltas) if deltas else None)
    print('PARITY', json.dumps(summary))
    try:
        os.makedirs('gpurun_out', exist_ok=True)
        with open(os.path.join('gpurun_out', f'parity_peak{peak_logit:g}_noise{noise_gain:g}.json'), 'w') as f:
            json.dump(summary, f, indent=1)
    except OSError:
        pass
    # 100 % wherever the fp32 engine's own decision margin exceeds the stated gap -- at every sharpness
    for row in table:
        if row['gap_lo'] >= GAP_CERTAIN and row['n']:
            assert row['agreement'] == 1.0, row
    for d in deltas:
        assert d <= 1e-4, d                     # north_star: 1e-4 relative on the reprojection error where indices agree
    if must_agree:                               # the bench workload's sharpness: every usable keypoint identical, cameras found
        assert summary['fp32_on_stamped_cell'] >= 0.99
        assert summary['index_agreement_usable'] == 1.0, summary
        assert frames_equal == B and summary['cameras_fp32'] == B
