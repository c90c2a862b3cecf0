"""GPU: the other BASELINE.json configurations as parity cases (C2 W32 480x270 batch 32 decode-only; C4 keypoint +
line networks joined through the line-intersection candidates; C5 1920x1080 shapes), plus the frame pipeline."""
import numpy as np
import pytest
import torch

from oracle import decode as od
from oracle import hrnet_ref as hr
from oracle import lines as ol
from oracle import solve as osolve
from oracle import synth

pytestmark = pytest.mark.gpu


def test_c2_w32_480x270_batch32_decode(sncal, cuda):
    """C2: HRNet-W32 (build-defined widths 32/64/128/256; the reference ships no w32 yaml) at 480x270, batch 32,
    heatmap argmax only.  Head is 136x240 while the stem is 135x240 -> stem interpolation path, unfused head."""
    cfg = hr.load_config('hrnet_w32')
    sd = hr.seeded_state_dict(cfg, 11, 2.0)
    x = hr.seeded_input(32, 270, 480, 12)
    net = sncal.HRNetHeatmap('hrnet_w32', dtype='fp32', device=cuda)
    net.load_state_dict(sd)
    heat, kp = net.forward(x.to(cuda), want_heat=True, decode_size=(540, 960))
    assert heat.shape == (32, 58, 136, 240) and kp.shape == (32, 57, 3)
    ref = hr.forward(sd, x[:2], cfg).numpy()                     # the oracle on 2 of the 32 frames (seconds on CPU)
    assert np.abs(heat[:2].cpu().numpy() - ref).max() <= 2e-4
    assert np.array_equal(kp[:2].cpu().numpy()[..., :2], od.keypoint_decode(ref, (540, 960))[..., :2])
    # size-independent property over the whole batch: the fused decode equals the oracle decode of our own heatmaps
    assert np.array_equal(kp.cpu().numpy(), od.keypoint_decode(heat.cpu().numpy(), (540, 960)))
    # bf16 engine on the same batch stays close
    netb = sncal.HRNetHeatmap('hrnet_w32', dtype='bf16', device=cuda)
    netb.load_state_dict(sd)
    hb, _ = netb.forward(x.to(cuda))
    # bf16 drift is not a parity claim (that is the fp32 engine above): mean |dlogp| within two bf16 ulps of the
    # typical |logp| ~ 10 (ulp 0.06; measured 0.066), isolated worst case below 1 (measured 0.53)
    db = np.abs(hb[:2].cpu().numpy() - ref)
    assert db.mean() < 0.12 and db.max() < 1.0
    # the fp32-class engine (every branch width of W32 on the 64 x 12 x 32 two-team tile, stem-interpolation head on the generic
    # split-arithmetic kernels): oracle indices, log-probabilities to 1e-3
    net3 = sncal.HRNetHeatmap('hrnet_w32', dtype='fp16x3', device=cuda)
    net3.load_state_dict(sd)
    h3, k3 = net3.forward(x.to(cuda), want_heat=True, decode_size=(540, 960))
    assert np.abs(h3[:2].cpu().numpy() - ref).max() <= 1e-3
    assert np.array_equal(k3[:2].cpu().numpy()[..., :2], od.keypoint_decode(ref, (540, 960))[..., :2])
    assert np.array_equal(k3.cpu().numpy()[..., :2], kp.cpu().numpy()[..., :2])          # all 32 frames: the fp32 engine's indices


def test_c4_keypoint_and_line_networks_joined(sncal, cuda):
    """C4 data flow on small nets: line net -> 2-peak decode -> slope/intercept -> intersections -> extra keypoint
    candidates for the solve (export_line_result.py:85-131 + prediction.py:105-124), all through the host mirror."""
    cfgl = hr.load_config('line_hrnet_w18')
    sdl = hr.seeded_state_dict(cfgl, 4, 4.0)
    x = hr.seeded_input(1, 64, 96, 5)
    lnet = sncal.HRNetHeatmap('line_hrnet_w18', dtype='fp32', device=cuda)
    lnet.load_state_dict(sdl)
    heat = lnet(x.to(cuda))[-1]
    dec = sncal.EHMPredictionTransform.mask_heat_points_gauss(heat, sigma=3)
    ref_heat = hr.forward(sdl, x, cfgl).numpy()
    ref_dec = od.line_decode(ref_heat, 3.0, 1.0)
    assert np.array_equal(dec.cpu().numpy()[..., :2], ref_dec[..., :2])
    lines_m, pts_m = sncal.lines.get_line_data(dec, scale=4, prob_thre=0.0)
    lines_o, _ = ol.get_line_data(ref_dec, scale=4, prob_thre=0.0)
    assert lines_m.keys() == lines_o.keys() and len(lines_m) > 0
    km = sncal.lines.lines_to_keypoints({k: v for k, v in lines_m.items() if v[0] is not None})
    ko = ol.lines_to_keypoints({k: v for k, v in lines_o.items() if v[0] is not None})
    assert km.keys() == ko.keys()
    for k in km:
        assert np.allclose(km[k], ko[k], rtol=1e-4, atol=1e-3)
    arr = sncal.lines.keypoints_to_array(km)
    assert arr.shape == (30, 3) and arr[:, 2].sum() == len(km)
    # the same chain on the device (sncal_lines_to_points): bit-identical to the host mirror's array
    dev = sncal.lines.lines_to_points_device(dec, scale=4, prob_thre=0.0).cpu().numpy()
    assert np.array_equal(dev[0].view(np.uint32), arr.view(np.uint32))
    # ... and inside the pipeline: keypoint net + line net on the same frames, line candidates straight into the solve
    cfgk = hr.load_config('hrnet_w18')
    knet = sncal.HRNetHeatmap('hrnet_w18', dtype='fp32', device=cuda)
    knet.load_state_dict(hr.seeded_state_dict(cfgk, 6, 4.0))
    cc = sncal.CameraCreator(sncal.PITCH_POINTS, conf_thresh=0.5, conf_threshs=[0.5, 0.35, 0.2], algorithm='iterative_voter',
                             max_rmse=55.0, max_rmse_rel=5.0, min_points=5, min_focal_length=10.0,
                             min_points_per_plane=6, min_points_for_refinement=6, reliable_thresh=57)
    kps = torch.from_numpy(np.stack([synth.synth_keypoints(720)[0]])).to(cuda)
    pipe = sncal.CalibrationPipeline(knet, cc, line_net=lnet)
    out = pipe.submit(x.to(cuda), extra_keypoints=kps)
    cams = pipe.cameras(out[1])
    direct = cc.records(cc.solve_device(out[0], torch.from_numpy(dev).to(cuda)))
    assert len(cams) == 1 and (cams[0] is None) == (direct[0].status == 0)


def test_c5_1080p_shapes(sncal, cuda):
    """C5 geometry: 1920x1080 input -> (B,58,540,960) heatmaps, decode grid step 1 px (transforms.py:234-235).
    Small net, fp32, checked against the oracle on one frame (the fp8 arithmetic of C5 is not built yet)."""
    cfg = hr.load_config('hrnet_w18')
    sd = hr.seeded_state_dict(cfg, 31, 4.0)
    x = hr.seeded_input(1, 1080, 1920, 32)
    net = sncal.HRNetHeatmap('hrnet_w18', dtype='fp32', device=cuda)
    net.load_state_dict(sd)
    heat, kp = net.forward(x.to(cuda), want_heat=True, decode_size=(540, 960))
    assert heat.shape == (1, 58, 540, 960)
    ref = hr.forward(sd, x, cfg).numpy()
    assert np.abs(heat.cpu().numpy() - ref).max() <= 2e-4
    assert np.array_equal(kp.cpu().numpy()[..., :2], od.keypoint_decode(ref, (540, 960))[..., :2])


def test_c5_w48_1080p_flagship(sncal, cuda):
    """C5 shapes on the flagship network: HRNet-W48 at 1920x1080.  fp32 engine vs the oracle on one frame (identical
    keypoint indices, |dlogp| <= 2e-4); bf16 engine (the fused BasicBlock / head / decode kernels at 270x480 branch
    maps and a 540x960 head) on a small batch: same keypoint cells as the fp32 engine up to the bf16 drift bound used
    for C3.  (C5's e4m3 arithmetic and its per-GPU batch: test_c5_per_gpu_share_... below and tests/test_fp8_gpu.py.)"""
    cfg = hr.load_config('hrnet_w48')
    sd = hr.seeded_state_dict(cfg, 1, 1.5)
    x = hr.seeded_input(2, 1080, 1920, 41)
    net32 = sncal.HRNetHeatmap('hrnet_w48', dtype='fp32', device=cuda)
    net32.load_state_dict(sd)
    heat, kp32 = net32.forward(x[:1].to(cuda), want_heat=True, decode_size=(1080, 1920))
    assert heat.shape == (1, 58, 540, 960)
    ref = hr.forward(sd, x[:1], cfg).numpy()
    assert np.abs(heat.cpu().numpy() - ref).max() <= 2e-4
    assert np.array_equal(kp32.cpu().numpy()[..., :2], od.keypoint_decode(ref, (1080, 1920))[..., :2])
    del heat, net32
    torch.cuda.empty_cache()
    net16 = sncal.HRNetHeatmap('hrnet_w48', dtype='bf16', device=cuda)
    net16.load_state_dict(sd)
    h16, k16 = net16.forward(x.to(cuda), want_heat=True, decode_size=(1080, 1920))
    _, k16b = net16.forward(x.to(cuda), want_heat=False, decode_size=(1080, 1920))
    assert torch.equal(k16, k16b)                                  # fused decode at 540x960
    d = np.abs(h16[:1].cpu().numpy() - ref)
    assert d.mean() < 0.12 and d.max() < 1.0, (d.mean(), d.max())
    del h16, net16
    torch.cuda.empty_cache()
    net3 = sncal.HRNetHeatmap('hrnet_w48', dtype='fp16x3', device=cuda)          # the fp32-class engine at C5's shapes: oracle indices
    net3.load_state_dict(sd)
    h3, k3 = net3.forward(x[:1].to(cuda), want_heat=True, decode_size=(1080, 1920))
    assert np.abs(h3.cpu().numpy() - ref).max() <= 1e-3
    assert np.array_equal(k3.cpu().numpy()[..., :2], od.keypoint_decode(ref, (1080, 1920))[..., :2])


def test_pipeline_overlapped_solve_matches_direct(sncal, cuda):
    """CalibrationPipeline (side-stream solve) returns the same cameras as the synchronous calls."""
    cfg = hr.load_config('hrnet_w18')
    net = sncal.HRNetHeatmap('hrnet_w18', dtype='bf16', device=cuda)
    net.load_state_dict(hr.seeded_state_dict(cfg, 9, 4.0))
    cc = sncal.CameraCreator(sncal.PITCH_POINTS, conf_thresh=0.5, conf_threshs=[0.5, 0.35, 0.2], algorithm='iterative_voter',
                             max_rmse=55.0, max_rmse_rel=5.0, min_points=5, min_focal_length=10.0,
                             min_points_per_plane=6, min_points_for_refinement=6, reliable_thresh=57)
    pipe = sncal.CalibrationPipeline(net, cc)
    x = hr.seeded_input(4, 135, 240, 10).to(cuda)
    kps = torch.from_numpy(np.stack([synth.synth_keypoints(s)[0] for s in range(700, 708)])).to(cuda)
    outs = [pipe.submit(x, extra_keypoints=kps) for _ in range(3)]
    cams = pipe.cameras(outs[-1][2])
    direct = cc.solve_batch(kps)
    for a, b in zip(cams, direct):
        assert (a is None) == (b is None)
        if a is not None:
            assert a.rmse == b.rmse and np.array_equal(a.rotation, b.rotation)
    _, k_direct = net.forward(x, want_heat=False, decode_size=(540, 960))
    assert torch.equal(outs[0][0], k_direct)
    oc = osolve.CameraCreatorOracle()
    o0 = oc(kps[0].cpu().numpy(), None)
    assert (o0 is None) == (cams[0] is None)


def test_c4_at_size_w48_keypoint_and_w48_line_networks(sncal, cuda):
    """C4 at its own size (export_line_result.py:85-131 -> prediction.py:105-124, 356-364): HRNet-W48 keypoint net and
    HRNet-W48 line net on the same 960x540 frames, through CalibrationPipeline(line_net=...): two-peak decode, line
    equations and the 30 intersection candidates stay on the device and feed the solve.
      * line candidates: bit-identical to the oracle's join (oracle/lines.py) applied to the line net's OWN heatmaps;
      * keypoints / cameras: the peaked workload (synth.py) drives the solve; records equal a direct solve call given the
        same keypoints and line points; cameras are found for the stamped frames;
      * the bf16 line net stays inside the bf16 drift bound of the fp32 engine on these frames."""
    import bench
    B = 4
    sd_k = sncal.synth.peaked_state_dict(bench.seeded_weights('hrnet_w48', seed=1))
    sd_l = bench.seeded_weights('line_hrnet_w48', seed=2)
    frames, expect = sncal.synth.stamped_frames(B, seed=77)
    x = torch.from_numpy(frames).to(cuda)
    knet = sncal.HRNetHeatmap('hrnet_w48', dtype='bf16', device=cuda)
    knet.load_state_dict(sd_k)
    lnet = sncal.HRNetHeatmap('line_hrnet_w48', dtype='bf16', device=cuda)
    lnet.load_state_dict(sd_l)
    cc = sncal.CameraCreator(sncal.PITCH_POINTS, conf_thresh=0.5, conf_threshs=[0.5, 0.35, 0.2], algorithm='iterative_voter',
                             max_rmse=55.0, max_rmse_rel=5.0, min_points=5, min_focal_length=10.0,
                             min_points_per_plane=6, min_points_for_refinement=6, reliable_thresh=57)
    pipe = sncal.CalibrationPipeline(knet, cc, line_net=lnet, line_sigma=3.0, line_scale=4, line_prob_thre=0.0)
    kp, rec = pipe.submit(x)[:2]
    pipe.join()
    torch.cuda.synchronize()
    # the line branch, step by step, on the line net's own heatmaps
    heat_l = lnet(x)[-1]
    assert heat_l.shape == (B, 23, 135, 240)
    peaks = sncal.EHMPredictionTransform.mask_heat_points_gauss(heat_l, sigma=3.0)
    ref_peaks = od.line_decode(heat_l.cpu().numpy(), 3.0, 1.0)
    assert np.array_equal(peaks.cpu().numpy()[..., :2], ref_peaks[..., :2])
    dev_pts = sncal.lines.lines_to_points_device(peaks, scale=4, prob_thre=0.0)
    arr = ol.keypoints_array(peaks.cpu().numpy(), scale=4, prob_thre=0.0)
    assert np.array_equal(dev_pts.cpu().numpy().view(np.uint32), arr.view(np.uint32))               # bit-identical candidates
    # the pipeline's records == a direct solve on the same keypoints + line points
    direct = cc.records(cc.solve_device(kp, dev_pts))
    got = cc.records(rec)
    for a, d in zip(got, direct):
        assert a.status == d.status and a.rmse == d.rmse and list(a.rotation) == list(d.rotation)
    vis = expect[..., 2] > 0
    assert float((kp.cpu().numpy()[..., :2] == expect[..., :2]).all(-1)[vis].mean()) >= 0.99
    assert sum(r.status != 0 for r in got) >= B - 1
    # fp32 line engine on one frame: the bf16 line net stays close (softmax probabilities)
    l32 = sncal.HRNetHeatmap('line_hrnet_w48', dtype='fp32', device=cuda)
    l32.load_state_dict(sd_l)
    h32 = l32(x[:1])[-1]
    assert float((h32 - heat_l[:1]).abs().max()) <= 6e-2


def test_c4_per_gpu_share_batch64_w48_keypoint_and_line_networks(sncal, cuda):
    """C4 at its per-GPU share (512 frames over 8 GPUs = 64 frames per GPU): CalibrationPipeline(HRNet-W48 keypoint net, HRNet-W48 line
    net) on 64 frames of 960x540 in the fp32-class engine.  Size-independent property: a frame's keypoints, line candidates and camera
    record do not depend on the batch it travels in (the same frames alone, in a batch of 4, give the same bytes); the device line
    join equals oracle/lines.py on the line net's own heatmaps for 2 of the frames; cameras are found for the stamped frames."""
    import bench
    B = 64
    sd_k = sncal.synth.peaked_state_dict(bench.seeded_weights('hrnet_w48', seed=1), deep=True)
    sd_l = bench.seeded_weights('line_hrnet_w48', seed=2)
    frames, expect = sncal.synth.stamped_frames(B, seed=91)
    x = torch.from_numpy(frames).to(cuda)
    knet = sncal.HRNetHeatmap('hrnet_w48', dtype='fp16x3', device=cuda)
    knet.load_state_dict(sd_k)
    lnet = sncal.HRNetHeatmap('line_hrnet_w48', dtype='fp16x3', device=cuda)
    lnet.load_state_dict(sd_l)
    cc = sncal.CameraCreator(sncal.PITCH_POINTS, **bench.SOLVER_KW)
    pipe = sncal.CalibrationPipeline(knet, cc, line_net=lnet, line_sigma=3.0, line_scale=4, line_prob_thre=0.0)
    kp, rec = pipe.submit(x)[:2]
    pipe.join()
    torch.cuda.synchronize()
    kp, rec = kp.clone(), rec.clone()
    assert kp.shape == (B, 57, 3)
    sel = [0, 17, 42, 63]
    kp4, rec4 = pipe.submit(x[sel].contiguous())[:2]
    pipe.join()
    torch.cuda.synchronize()
    assert torch.equal(kp4, kp[sel]) and torch.equal(rec4, rec[sel])                      # batch independence, byte for byte
    heat_l = lnet(x[:2])[-1]
    peaks = sncal.EHMPredictionTransform.mask_heat_points_gauss(heat_l, sigma=3.0)
    assert np.array_equal(peaks.cpu().numpy()[..., :2], od.line_decode(heat_l.cpu().numpy(), 3.0, 1.0)[..., :2])
    dev_pts = sncal.lines.lines_to_points_device(peaks, scale=4, prob_thre=0.0)
    arr = ol.keypoints_array(peaks.cpu().numpy(), scale=4, prob_thre=0.0)
    assert np.array_equal(dev_pts.cpu().numpy().view(np.uint32), arr.view(np.uint32))     # bit-identical candidates
    got = cc.records(rec)
    assert sum(r.status != 0 for r in got) >= B - 4
    vis = expect[..., 2] > 0
    near = np.abs(kp.cpu().numpy()[..., :2] - expect[..., :2]).max(-1) <= 8.0
    assert float(near[vis & (kp.cpu().numpy()[..., 2] >= 0.2)].mean()) >= 0.98


@pytest.mark.parametrize('dtype', ['fp16x3', 'fp8'])
def test_c5_per_gpu_share_batch128_w48_1080p(sncal, cuda, dtype):
    """C5 at its per-GPU share (1024 frames over 8 GPUs = 128 frames per GPU): HRNet-W48 on 128 frames of 1920x1080 (two sub-batches
    of 64), fp32-class engine and e4m3 engine.  Size-independent property: a frame's decoded keypoints do not depend on the batch it
    travels in (frames alone = the same bytes); the fp32-class engine reproduces the oracle's indices on one frame (the e4m3 engine
    is a tolerance study: most usable rows, tests/test_fp8_gpu.py)."""
    cfg = hr.load_config('hrnet_w48')
    sd = hr.seeded_state_dict(cfg, 1, 1.5)
    B = 128
    g = torch.Generator(device=cuda)
    g.manual_seed(505)
    x = torch.rand((B, 3, 1080, 1920), device=cuda, generator=g)
    net = sncal.HRNetHeatmap('hrnet_w48', dtype=dtype, device=cuda)
    net.load_state_dict(sd)
    if dtype == 'fp8':
        net.calibrate_fp8(x[:8])
    _, kp = net.forward(x, want_heat=False, decode_size=(1080, 1920))
    kp = kp.clone()
    assert kp.shape == (B, 57, 3) and bool(torch.isfinite(kp).all())
    sel = [0, 63, 64, 127]                                                                # both sub-batches, their edges
    _, kps = net.forward(x[sel].contiguous(), want_heat=False, decode_size=(1080, 1920))
    assert torch.equal(kps, kp[sel])
    if dtype == 'fp16x3':
        ref = hr.forward(sd, x[:1].cpu(), cfg).numpy()
        assert np.array_equal(kp[:1].cpu().numpy()[..., :2], od.keypoint_decode(ref, (1080, 1920))[..., :2])
