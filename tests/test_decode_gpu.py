"""GPU: HIP decodes (through the C ABI) vs the oracle and the reference-captured goldens.
Bar: indices bit-exact; conf bit-exact vs the oracle (same exp definition), 1 ULP vs the reference."""
import os

import numpy as np
import pytest
import torch

from oracle import decode as od
from oracle import synth
from test_oracle_goldens import GAUSS_CASES, regen_gauss_case

pytestmark = pytest.mark.gpu


def _dec(sncal, lp, cuda, size=(540, 960)):
    t = sncal.HRNetPredictionTransform(size)
    return t(torch.from_numpy(lp).to(cuda)).cpu().numpy()


@pytest.mark.parametrize('case', ['gauss_68x120', 'gauss_135x240_neginf', 'ties_34x60', 'noise_34x60'])
def test_keypoint_decode_golden(sncal, cuda, gold_dir, case):
    g = np.load(os.path.join(gold_dir, 'decode_keypoints.npz'))
    lp = g[case + '.in']
    if lp.size == 0:
        lp = regen_gauss_case(g, case)
    out = _dec(sncal, lp, cuda)
    ref = g[case + '.out']
    assert np.array_equal(out[..., :2], ref[..., :2])
    assert np.allclose(out[..., 2], ref[..., 2], rtol=2e-7, atol=0)
    assert np.array_equal(out, od.keypoint_decode(lp, (540, 960)))      # bit-exact vs the oracle


@pytest.mark.parametrize('hw', [(270, 480), (136, 240), (17, 30), (540, 960), (33, 61), (5, 3)])
def test_keypoint_decode_shapes_vs_oracle(sncal, cuda, hw):
    """C3 / C2 / odd / C5 sizes incl. widths that are not multiples of 4 (scalar path)."""
    rng = np.random.Generator(np.random.PCG64(hw[0] * 1000 + hw[1]))
    B, C = 2, 58
    lp = torch.log_softmax(torch.from_numpy((rng.random((B, C) + hw) * 12).astype(np.float32)), 1).numpy()
    # plant exact ties and saturated peaks
    lp[0, 3, hw[0] // 2, hw[1] // 3] = lp[0, 3].max() + 1
    lp[0, 3, hw[0] // 3, hw[1] // 2] = lp[0, 3, hw[0] // 2, hw[1] // 3]
    lp[1, 7, 1:3, 1:3] = -1e-9
    out = _dec(sncal, lp, cuda)
    assert np.array_equal(out, od.keypoint_decode(lp, (540, 960)))


def test_keypoint_decode_roundtrip_full_size(sncal, cuda):
    """BASELINE config C3 size (batch 8 slice): synthetic cameras -> heatmaps -> decode recovers the
    planted grid points exactly (size-independent property; the oracle also checks 2 frames)."""
    seeds = list(range(8))
    lp, kps = synth.synth_logp(seeds, hw=(270, 480))
    out = _dec(sncal, lp, cuda)
    vis = kps[..., 2] > 0.5
    assert np.array_equal(out[vis][:, :2], kps[vis][:, :2])
    assert np.allclose(out[vis][:, 2], kps[vis][:, 2], rtol=1e-5)
    assert (out[~vis][:, 2] < 1e-6).all()
    assert np.array_equal(out[:2], od.keypoint_decode(lp[:2], (540, 960)))


def test_empty_batch_and_errors(sncal, cuda):
    t = sncal.HRNetPredictionTransform((540, 960))
    assert t(torch.empty((0, 58, 8, 8), device=cuda)).shape == (0, 57, 3)
    with pytest.raises(sncal._lib.SncalError):
        t(torch.zeros((1, 58, 8, 8)))                       # CPU tensor: no CPU path
    with pytest.raises(sncal._lib.SncalError):
        t(torch.zeros((1, 58, 8, 8), device=cuda, dtype=torch.float64))


@pytest.mark.parametrize('sigma', [3, 6])
def test_line_decode_golden(sncal, cuda, gold_dir, sigma):
    g = np.load(os.path.join(gold_dir, 'decode_lines.npz'))
    t = sncal.EHMPredictionTransform(scale=4, sigma=sigma)
    out = t(torch.from_numpy(g['heat']).to(cuda)).cpu().numpy()
    ref = g[f'out_sigma{sigma}']
    assert np.array_equal(out[..., :2], ref[..., :2])
    assert np.allclose(out[..., 2], ref[..., 2], rtol=1e-5, atol=1e-7)
    assert np.array_equal(out, od.line_decode(g['heat'], float(sigma), 4.0))


def test_line_decode_full_size_vs_oracle(sncal, cuda):
    rng = np.random.Generator(np.random.PCG64(77))
    heat = (rng.random((1, 23, 135, 240)).astype(np.float32) - 0.3)
    heat[0, 2] = 0
    out = sncal.EHMPredictionTransform(scale=4, sigma=3)(torch.from_numpy(heat).to(cuda)).cpu().numpy()
    assert np.array_equal(out, od.line_decode(heat, 3.0, 4.0))
    raw = sncal.EHMPredictionTransform.mask_heat_points_gauss(torch.from_numpy(heat).to(cuda), sigma=3).cpu().numpy()
    assert np.array_equal(raw, od.line_decode(heat, 3.0, 1.0))


def test_lines_to_points_matches_oracle_bit_exact(sncal, cuda, gold_dir):
    """sncal_lines_to_points (L3 + L4 on the device) vs oracle.lines.keypoints_array: float32 outputs bit-identical
    (both run the float64 arithmetic of the reference's pinned numpy on float32 peaks)."""
    from oracle import lines as ol
    from sncal_amd.lines import lines_to_points_device
    rng = np.random.default_rng(11)
    B = 96
    peaks = np.zeros((B, 23, 2, 3), dtype=np.float32)
    peaks[..., 0] = rng.integers(0, 240, size=(B, 23, 2))
    peaks[..., 1] = rng.integers(0, 135, size=(B, 23, 2))
    peaks[..., 2] = rng.uniform(0.0, 1.0, size=(B, 23, 2))
    peaks[0, :, :, 2] = 0.2                       # p == threshold counts (>=)
    peaks[1, :, 1] = peaks[1, :, 0]               # identical peaks -> the reference's (None, None) line -> no line
    peaks[2, :, :, 1] = 50.0                      # all lines horizontal: |k1 - k2| <= 1e-4 -> no intersections
    peaks[3, :, 1, 0] = peaks[3, :, 0, 0]         # vertical lines: slope through the 1e-5 delta
    peaks[3, :, 1, 1] = peaks[3, :, 0, 1] + 7
    g = np.load(os.path.join(gold_dir, 'decode_lines.npz'))
    peaks[4] = g['out_sigma3'][0] / np.array([4, 4, 1], dtype=np.float32)      # the reference-captured decode
    ref = ol.keypoints_array(peaks, scale=4, prob_thre=0.2)
    got = lines_to_points_device(torch.from_numpy(peaks).to(cuda), scale=4.0, prob_thre=0.2).cpu().numpy()
    assert np.array_equal(got[..., 2], ref[..., 2])
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))
    assert ref[2, :, 2].sum() == 0 and ref[1, :, 2].sum() == 0 and ref[4, :, 2].sum() >= 10
    empty = lines_to_points_device(torch.zeros((0, 23, 2, 3), device=cuda))
    assert empty.shape == (0, 30, 3)
