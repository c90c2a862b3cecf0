"""GPU: the frame pipeline's pool of solve streams (pipeline.py, round 5) at the REFERENCE's refine criterion
(baseline/camera.py:116: solvePnPRefineLM criteria (20000, 1e-5) = the library default).

The pipeline solves batch k on solve stream k mod P (CU-masked streams when the caller stays off the null stream) while the network
runs on; what it returns must be, byte for byte and in submission order, what the synchronous CameraCreator.solve_device call returns
for the same keypoints."""
import numpy as np
import pytest
import torch

from oracle import hrnet_ref as hr

pytestmark = pytest.mark.gpu

KW = dict(conf_thresh=0.5, conf_threshs=[0.5, 0.35, 0.2], algorithm='iterative_voter', max_rmse=55.0, max_rmse_rel=5.0,
          min_points=5, min_focal_length=10.0, min_points_per_plane=6, min_points_for_refinement=6, reliable_thresh=57)


def noisy_keypoints(sncal, n, seed0=0):
    """n synthetic frames of keypoints, noise 0.5 / 1 / 2 / 4 px in turn: the 4-px ones hold the slow Levenberg-Marquardt fits."""
    rows = []
    for s in range(n):
        rng = np.random.default_rng(seed0 + s)
        cam = sncal.synth.random_camera(rng)
        rows.append(sncal.synth.keypoints_for_camera(cam, rng, sigma_px=(0.5, 1.0, 2.0, 4.0)[s % 4]))
    return np.stack(rows).astype(np.float32)


def test_pool_records_equal_the_synchronous_solve_in_submission_order(sncal, cuda):
    cfg = hr.load_config('hrnet_w18')
    sd = hr.seeded_state_dict(cfg, 3, 4.0)
    net = sncal.HRNetHeatmap('hrnet_w18', dtype='fp16x3', device=cuda)
    net.load_state_dict(sd)
    cc = sncal.CameraCreator(sncal.PITCH_POINTS, **KW)
    assert cc.refine_max_iters == 20000                       # the reference's criterion, not the cap of rounds 1-4
    pipe = sncal.CalibrationPipeline(net, cc, decode_size=(540, 960))
    n_streams = pipe.max_in_flight // 2
    assert n_streams >= 2, 'the pool is what this test is about (SNCAL_SOLVE_STREAMS)'
    x = hr.seeded_input(8, 135, 240, 4).to(cuda)
    nb, per = 3 * n_streams + 1, 48                           # more batches than streams and than the in-flight bound
    kp = torch.from_numpy(noisy_keypoints(sncal, nb * per)).to(cuda)
    outs = []
    torch.cuda.synchronize()
    with pipe.stream():                                       # off the null stream: the CU-masked solve streams (pipeline.py)
        for b in range(nb):
            outs.append(pipe.submit(x, extra_keypoints=kp[b * per:(b + 1) * per].contiguous()))
        assert pipe.masked == (sncal.pipeline.SOLVE_CUS_PER_XCD > 0)
        pipe.join()
    torch.cuda.synchronize()
    assert len(pipe._pending) == 0
    found = 0
    for b, (k_dec, rec_dec, rec_extra) in enumerate(outs):
        ref_extra = cc.solve_device(kp[b * per:(b + 1) * per].contiguous())
        ref_dec = cc.solve_device(k_dec)
        torch.cuda.synchronize()
        assert torch.equal(rec_extra, ref_extra), f'batch {b}: pooled solve differs from the synchronous call'
        assert torch.equal(rec_dec, ref_dec), f'batch {b}: pooled solve of the decoded keypoints differs'
        found += sum(r.status != 0 for r in cc.records(rec_extra))
    assert found >= 0.8 * nb * per                            # the noisy frames are solvable: the comparison is not None == None


def test_cameras_follow_submission_order(sncal, cuda):
    """cameras() of an EARLIER batch after later ones were submitted: in-order delivery, not stream order."""
    cfg = hr.load_config('hrnet_w18')
    sd = hr.seeded_state_dict(cfg, 3, 4.0)
    net = sncal.HRNetHeatmap('hrnet_w18', dtype='fp16x3', device=cuda)
    net.load_state_dict(sd)
    cc = sncal.CameraCreator(sncal.PITCH_POINTS, **KW)
    pipe = sncal.CalibrationPipeline(net, cc, decode_size=(540, 960))
    x = hr.seeded_input(2, 135, 240, 4).to(cuda)
    kp = torch.from_numpy(noisy_keypoints(sncal, 5 * 16, seed0=500)).to(cuda)
    outs = [pipe.submit(x, extra_keypoints=kp[b * 16:(b + 1) * 16].contiguous()) for b in range(5)]
    assert not pipe.masked                                   # a caller on the null stream gets plain non-blocking solve streams
    cams0 = pipe.cameras(outs[0][2])
    ref0 = cc.solve_batch(kp[:16])
    assert [c is None for c in cams0] == [c is None for c in ref0]
    for a, b in zip(cams0, ref0):
        if a is not None:
            assert a.rmse == b.rmse and np.array_equal(a.position, b.position)


def test_line_branch_records_equal_the_synchronous_pieces(sncal, cuda):
    """C4's data flow through the pipeline (keypoint net + line net + two-peak decode + device line join + solve WITH line points on the
    pooled CU-masked streams) against the same pieces called one by one: the records must be the bytes of
    CameraCreator.solve_device(kpts, line_points) on the pipeline's own keypoints and line points."""
    import bench
    from sncal_amd.lines import lines_to_points_device
    B = 4
    knet = sncal.HRNetHeatmap('hrnet_w48', dtype='fp16x3', device=cuda)
    knet.load_state_dict(sncal.synth.peaked_state_dict(bench.seeded_weights('hrnet_w48', seed=1), deep=True))
    lnet = sncal.HRNetHeatmap('line_hrnet_w48', dtype='fp16x3', device=cuda)
    lnet.load_state_dict(sncal.synth.line_deep_state_dict(bench.seeded_weights('line_hrnet_w48', seed=2)))
    cc = sncal.CameraCreator(sncal.PITCH_POINTS, **KW)
    frames, _ = sncal.synth.stamped_frames(3 * B, seed=1000, size=(540, 960))
    x = torch.from_numpy(frames).to(cuda)
    for thre in (0.2, 0.0):                                   # the designed join, and the export CLI's prob_thre 0 (every line pair "intersects")
        pipe = sncal.CalibrationPipeline(knet, cc, decode_size=(540, 960), line_net=lnet, line_prob_thre=thre)
        with pipe.stream():
            outs = [pipe.submit(x[b * B:(b + 1) * B]) for b in range(3)]
            pipe.join()
        torch.cuda.synchronize()
        n_lp = 0
        for b, (kpts, rec) in enumerate(outs):
            xb = x[b * B:(b + 1) * B]
            _, k_ref = knet.forward(xb, want_heat=False, decode_size=(540, 960))
            heat, _ = lnet.forward(xb, want_heat=True)
            lp = lines_to_points_device(sncal.EHMPredictionTransform.mask_heat_points_gauss(heat, sigma=3.0), scale=4.0, prob_thre=thre)
            ref = cc.solve_device(k_ref, lp)
            torch.cuda.synchronize()
            assert torch.equal(kpts, k_ref) and torch.equal(rec, ref), (thre, b)
            n_lp += int((lp[..., 2] > 0.5).sum())
        assert n_lp > 0
