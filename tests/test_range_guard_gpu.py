"""GPU: the fp16 range guard of the default engine (fp16x3: every operand as fp16 hi + lo halves, clamped at +-65504; x3.hpp).

The reference's predict() is plain fp32 with no range limit (src/models/hrnet/metamodel.py:127-134), so the engine must never hand back
plausible heatmaps from a checkpoint or an input it cannot represent:
  * load time   sncal_hrnet_finalize refuses folded weights beyond the fp16 range / below its normal range (SNCAL_ERR_RANGE); load_model
                falls back to the exact-fp32 engine and says so;
  * run time    every kernel tracks the activations it produces; sncal_hrnet_range_status reports clamped values and non-finite inputs;
  * in between  the engine is scale-robust where the scale is free: HRNetHeatmap.load_state_dict rebalances block-internal channels by
                exact powers of two (_equalize_blocks), so rescaling a BatchNorm's affine by 2^k and the next convolution by 2^-k -- bit-identical
                in the fp32 engine -- keeps the fp32 indices within fp16x3's golden tolerance; where it cannot, the engine refuses or raises the flag.
"""
import warnings

import numpy as np
import pytest
import torch

from oracle import hrnet_ref as hr

pytestmark = pytest.mark.gpu
TOL_LOGP = 2e-4            # the golden tolerance on log-probabilities, fp32 and fp16x3 engines alike (tests/test_hrnet_gpu.py)


def _nets(sncal, cuda, sd, dtype):
    net = sncal.HRNetHeatmap('hrnet_w18', dtype=dtype, device=cuda)
    net.load_state_dict(sd)
    return net


def test_finalize_refuses_what_fp16_halves_cannot_hold_and_load_model_falls_back(sncal, cuda, tmp_path):
    cfg = hr.load_config('hrnet_w18')
    sd = hr.seeded_state_dict(cfg, 5, 4.0)
    big = {k: v.clone() for k, v in sd.items()}
    bn = 'model.stage3.0.branches.1.1.bn2'                  # (bn2 closes the block: its output joins the residual stream, nothing to rebalance)
    big[bn + '.running_var'][3] = 1e-12                     # a near-dead channel: scale = gamma / sqrt(eps) = 316 gamma
    big[bn + '.weight'][3] = 4000.0
    with pytest.raises(sncal._lib.SncalRangeError, match='stage3.0.branches.1.1.conv2'):
        _nets(sncal, cuda, big, 'fp16x3')
    _nets(sncal, cuda, big, 'fp32')                         # the reference's arithmetic takes it
    tiny = {k: v.clone() for k, v in sd.items()}
    tiny['model.transition1.0.0.weight'] *= 2.0 ** -16     # (not a block-internal pair: load_state_dict cannot rebalance it away)
    with pytest.raises(sncal._lib.SncalRangeError, match='below 2\\^-14'):
        _nets(sncal, cuda, tiny, 'fp16x3')
    nan = {k: v.clone() for k, v in sd.items()}
    nan['model.conv2.weight'][0, 0, 0, 0] = float('nan')
    with pytest.raises(sncal._lib.SncalRangeError):
        _nets(sncal, cuda, nan, 'fp16x3')
    # the drop-in default falls back to fp32 (with a warning); an engine asked for by name does not
    ck = {'model_name': 'HRNetMetaModel',
          'params': {'nn_module': {'hrnet_config': cfg, 'num_refinement_stages': 0, 'num_heatmaps': 58},
                     'prediction_transform': {'size': [540, 960]}, 'device': 'cuda:0'},
          'nn_state_dict': big}
    path = str(tmp_path / 'big.pth')
    torch.save(ck, path)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        model = sncal.load_model(path, device='cuda:0')
    assert model.nn_module.dtype_name == 'fp32' and any('falling back' in str(m.message) for m in w)
    with pytest.raises(sncal._lib.SncalRangeError):
        sncal.load_model(path, device='cuda:0', dtype='fp16x3')
    x = hr.seeded_input(1, 135, 240, 6)
    assert model.predict(x).shape == (1, 57, 3)


def test_run_time_flag_counts_clamped_activations_and_nonfinite_inputs(sncal, cuda):
    cfg = hr.load_config('hrnet_w18')
    sd = hr.seeded_state_dict(cfg, 5, 4.0)
    x = hr.seeded_input(2, 135, 240, 6).to(cuda)
    net = _nets(sncal, cuda, sd, 'fp16x3')
    net.forward(x, want_heat=False, decode_size=(540, 960))
    assert net.range_status() == (0, 0)
    # activations beyond 65504 with weights that pass the load-time check: the stem's BatchNorm affine x 2^15
    hot = {k: v.clone() for k, v in sd.items()}
    hot['model.bn1.weight'] *= 2.0 ** 15
    hot['model.bn1.bias'] *= 2.0 ** 15
    hot['model.conv2.weight'] *= 2.0 ** -6                 # (keeps the rest of the network finite in fp32)
    nh = _nets(sncal, cuda, hot, 'fp16x3')
    nh.forward(x, want_heat=False, decode_size=(540, 960))
    ov, nf = nh.range_status(clear=False)
    assert ov > 0 and nf == 0
    with pytest.raises(sncal._lib.SncalRangeError, match='beyond the fp16 range'):
        nh.range_status(check=True)
    assert nh.range_status() == (0, 0)                      # cleared
    n32 = _nets(sncal, cuda, hot, 'fp32')
    h32, _ = n32.forward(x, want_heat=True)
    assert torch.isfinite(h32).all() and n32.range_status() == (0, 0)
    # a NaN in the frame
    xb = x.clone()
    xb[1, 2, 17, 33] = float('nan')
    net.forward(xb, want_heat=False, decode_size=(540, 960))
    ov, nf = net.range_status()
    assert nf > 0
    # a FINITE frame value beyond 65504 (forward() takes any fp32 tensor, not only ToTensor's [0, 1]): the stem's in-kernel split would
    # clamp it silently -- the input layout kernel counts it as an overflow
    xh = x.clone()
    xh[0, 1, 40, 80] = 7.0e4
    net.forward(xh, want_heat=False, decode_size=(540, 960))
    ov, nf = net.range_status()
    assert ov > 0 and nf == 0
    n32f = _nets(sncal, cuda, sd, 'fp32')
    n32f.forward(xh, want_heat=False, decode_size=(540, 960))
    assert n32f.range_status() == (0, 0)


def test_predict_raises_on_the_offending_call_not_one_call_late(sncal, cuda, tmp_path):
    """predict() used to check the flag of the calls BEFORE it, so the last or only predict() of a run was never
    checked.  The drop-in surface is load_model(...).predict(x) (make_submit.py:51,68)."""
    cfg = hr.load_config('hrnet_w18')
    sd = hr.seeded_state_dict(cfg, 5, 4.0)
    hot = {k: v.clone() for k, v in sd.items()}
    hot['model.bn1.weight'] *= 2.0 ** 15
    hot['model.bn1.bias'] *= 2.0 ** 15
    hot['model.conv2.weight'] *= 2.0 ** -6
    params = {'nn_module': {'hrnet_config': cfg, 'num_refinement_stages': 0, 'num_heatmaps': 58},
              'prediction_transform': {'size': [540, 960]}, 'device': 'cuda:0'}
    x = hr.seeded_input(2, 135, 240, 6)
    path = str(tmp_path / 'hot.pth')
    torch.save({'model_name': 'HRNetMetaModel', 'params': params, 'nn_state_dict': hot}, path)
    model = sncal.load_model(path, device='cuda:0', dtype='fp16x3')
    with pytest.raises(sncal._lib.SncalRangeError):
        model.predict(x)                                    # the FIRST and only call
    assert model.predict(x, check_range=False).shape == (2, 57, 3)      # opt-out: no wait, the caller owes the check
    with pytest.raises(sncal._lib.SncalRangeError):
        model.check_range()
    path = str(tmp_path / 'ok.pth')
    torch.save({'model_name': 'HRNetMetaModel', 'params': params, 'nn_state_dict': sd}, path)
    good = sncal.load_model(path, device='cuda:0')
    assert good.predict(x).shape == (2, 57, 3) and good.predict(x).shape == (2, 57, 3)


def test_pipeline_and_predict_surface_raise_on_the_flag(sncal, cuda):
    cfg = hr.load_config('hrnet_w18')
    sd = hr.seeded_state_dict(cfg, 5, 4.0)
    hot = {k: v.clone() for k, v in sd.items()}
    hot['model.bn1.weight'] *= 2.0 ** 15
    hot['model.bn1.bias'] *= 2.0 ** 15
    hot['model.conv2.weight'] *= 2.0 ** -6
    net = _nets(sncal, cuda, hot, 'fp16x3')
    cc = sncal.CameraCreator(sncal.PITCH_POINTS, algorithm='iterative_voter')
    pipe = sncal.CalibrationPipeline(net, cc, decode_size=(540, 960))
    out = pipe.submit(hr.seeded_input(2, 135, 240, 6).to(cuda))
    with pytest.raises(sncal._lib.SncalRangeError):
        pipe.cameras(out[1])


@pytest.mark.parametrize('where', ['stage2.0.branches.0.1', 'stage3.0.branches.2.0', 'layer1.0'])
def test_scale_equivariance_of_a_block(sncal, cuda, where):
    """bn1's affine x 2^k, the next convolution's input columns x 2^-k, k in [-14, 14]: the fp32 engine's output is bit-identical to
    k = 0; fp16x3 keeps the fp32 indices and its golden tolerance, or refuses the checkpoint, or raises the range flag -- never a
    silently different heatmap."""
    cfg = hr.load_config('hrnet_w18')
    sd = hr.seeded_state_dict(cfg, 9, 4.0)
    x = hr.seeded_input(2, 135, 240, 10).to(cuda)
    units = sncal.HRNetHeatmap('hrnet_w18', dtype='fp32', device='cpu').conv_units()
    n32 = _nets(sncal, cuda, sd, 'fp32')
    h0, k0 = n32.forward(x, want_heat=True, decode_size=(540, 960))
    outcomes = {}
    for k in range(-14, 15, 2):
        sdk = sncal.synth.rescaled_state_dict(sd, units, only=where, k_fixed=k)
        assert any(not torch.equal(sdk[n], sd[n]) for n in sd) or k == 0
        nk = _nets(sncal, cuda, sdk, 'fp32')
        hk, kk = nk.forward(x, want_heat=True, decode_size=(540, 960))
        assert torch.equal(hk, h0) and torch.equal(kk, k0), f'fp32 engine is not invariant at k = {k}'
        try:
            n3 = _nets(sncal, cuda, sdk, 'fp16x3')
        except sncal._lib.SncalRangeError:
            outcomes[k] = 'refused at load'
            continue
        h3, k3 = n3.forward(x, want_heat=True, decode_size=(540, 960))
        ov, nf = n3.range_status()
        if ov or nf:
            outcomes[k] = 'range flag'
            continue
        err = float((h3 - h0).abs().max())
        assert torch.equal(k3[..., :2], k0[..., :2]), f'k = {k}: keypoint indices differ from the fp32 engine'
        assert err <= TOL_LOGP, f'k = {k}: |dlogp| {err:.2e} beyond the golden tolerance without a refusal or a flag'
        outcomes[k] = f'ok {err:.1e}'
    print('SCALE-EQUIVARIANCE', where, outcomes)
    assert all(v.startswith('ok') for v in outcomes.values()), outcomes          # block-internal scales are rebalanced away at load


def test_trained_like_scales_keep_the_fp32_indices(sncal, cuda):
    """A checkpoint whose block-internal channels span four decades (log-normal power-of-two scales, a few near-dead channels;
    synth.rescaled_state_dict): same function bit for bit in fp32, and the fp16x3 engine returns the fp32 engine's keypoint indices
    within its tolerance -- or says that it cannot."""
    cfg = hr.load_config('hrnet_w18')
    sd = hr.seeded_state_dict(cfg, 9, 4.0)
    x = hr.seeded_input(4, 135, 240, 10).to(cuda)
    units = sncal.HRNetHeatmap('hrnet_w18', dtype='fp32', device='cpu').conv_units()
    n32 = _nets(sncal, cuda, sd, 'fp32')
    h0, k0 = n32.forward(x, want_heat=True, decode_size=(540, 960))
    done = 0
    for seed in range(4):
        sdk = sncal.synth.rescaled_state_dict(sd, units, seed=seed, sigma_log2=3.0, dead_frac=0.02)
        hk, kk = _nets(sncal, cuda, sdk, 'fp32').forward(x, want_heat=True, decode_size=(540, 960))
        assert torch.equal(hk, h0) and torch.equal(kk, k0)
        try:
            n3 = _nets(sncal, cuda, sdk, 'fp16x3')
        except sncal._lib.SncalRangeError:
            continue
        h3, k3 = n3.forward(x, want_heat=True, decode_size=(540, 960))
        if n3.range_status() != (0, 0):
            continue
        assert n3.equalized > 100                          # the four decades were taken out at load
        assert torch.equal(k3[..., :2], k0[..., :2]) and float((h3 - h0).abs().max()) <= TOL_LOGP
        done += 1
    assert done == 4, 'block-internal scales are rebalanced at load: every such checkpoint must run'
