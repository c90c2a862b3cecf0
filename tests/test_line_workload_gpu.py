"""GPU: the C4 bench's designed line network (synth.line_deep_state_dict) through the product path -- line net forward (Softmax head,
src/models/line/hrnet.py:86-102), two-peak decode (src/models/line/transforms.py:217-280), device line join (export_line_result.py:51-131,
prediction.py:105-124) -- gives line points that are CONSISTENT with the stamped frame: a line point is valid only where both lines of
its pair show two stamped keypoints, and it lands on the keypoint those lines cross in."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_designed_line_network_yields_line_points_on_the_stamped_keypoints(sncal, cuda):
    import bench
    from sncal_amd.lines import lines_to_points_device
    B = 4
    sdl = sncal.synth.line_deep_state_dict(bench.seeded_weights('line_hrnet_w48', seed=2))
    ln = sncal.HRNetHeatmap('line_hrnet_w48', dtype='fp16x3', device=cuda)
    ln.load_state_dict(sdl)
    frames, expect = sncal.synth.stamped_frames(B, seed=1000, size=(540, 960))
    heat, _ = ln.forward(torch.from_numpy(frames).to(cuda), want_heat=True)
    assert heat.shape == (B, 23, 135, 240)
    peaks = sncal.EHMPredictionTransform.mask_heat_points_gauss(heat, sigma=3.0)
    lp = lines_to_points_device(peaks, scale=4.0, prob_thre=0.2).cpu().numpy()          # (B,30,3) [x, y, valid]
    on = sncal.synth.line_keypoints()
    n_valid, errs = 0, []
    for b in range(B):
        vis = expect[b, :, 2] > 0
        shows_two = {l: sum(bool(vis[k]) for k in kps) >= 2 for l, kps in on.items()}
        for k in range(30):
            if lp[b, k, 2] > 0.5:
                n_valid += 1
                assert vis[k], (b, k, 'a line point where no keypoint was stamped')
                lines_k = [l for l, kps in on.items() if k in kps]
                assert all(shows_two[l] for l in lines_k), (b, k)
                errs.append(float(np.linalg.norm(lp[b, k, :2] - expect[b, k, :2])))
    assert n_valid >= 2 * B                                    # and there ARE line points
    # peaks sit on the 4 px heat grid, a line goes through two of them: its crossing with another such line is a few pixels off the
    # keypoint, more where a line's two peaks are close together (what a trained line network's output looks like as well)
    assert np.median(errs) <= 8.0 and max(errs) <= 48.0, sorted(errs)[-5:]
    # the raw random line network at the export CLI's prob_thre 0 (the round 2-5 C4 workload): every pair of lines "intersects" somewhere
    lr = sncal.HRNetHeatmap('line_hrnet_w48', dtype='fp16x3', device=cuda)
    lr.load_state_dict(bench.seeded_weights('line_hrnet_w48', seed=2))
    heat_r, _ = lr.forward(torch.from_numpy(frames).to(cuda), want_heat=True)
    lp_r = lines_to_points_device(sncal.EHMPredictionTransform.mask_heat_points_gauss(heat_r, sigma=3.0), scale=4.0, prob_thre=0.0).cpu().numpy()
    assert (lp_r[..., 2] > 0.5).sum() >= 25 * B
