"""GPU: sncal_create_target (training-target synthesis) vs the oracle (bit-exact) and the reference capture."""
import os

import numpy as np
import pytest
import torch

from oracle import synth
from test_oracle_goldens import _target_close

pytestmark = pytest.mark.gpu


def test_target_equals_oracle_and_reference_capture(sncal, cuda, gold_dir):
    g = np.load(os.path.join(gold_dir, 'target.npz'))
    for name in ('small', 'train'):
        kp, sigma, hw = g[f'{name}.kp'], float(g[f'{name}.sigma']), tuple(int(v) for v in g[f'{name}.hw'])
        got = sncal.loss.create_target(torch.from_numpy(kp).to(cuda), sigma, hw).cpu().numpy()
        want = synth.create_target(kp, sigma, hw)
        assert got.shape == want.shape and np.array_equal(got, want), name         # same arithmetic, correctly rounded exp
        if name == 'small':
            assert _target_close(g['small.target'], got, kp.shape[1])
        else:
            assert np.allclose(got.astype(np.float64).sum(axis=(2, 3)), g['train.chan_sum'], rtol=1e-6, atol=1e-4)
    # create_heatmaps mirror: (B,N,2) input, no background channel
    kp = g['small.kp']
    hm = sncal.loss.create_heatmaps(torch.from_numpy(kp[..., :2].copy()).to(cuda), 2.0, (20, 33)).cpu().numpy()
    kp2 = kp.copy(); kp2[..., 2] = 0
    assert np.array_equal(hm, synth.create_target(kp2, 2.0, (20, 33))[:, :-1])


def test_target_edge_shapes(sncal, cuda):
    """Ragged sizes: width not a multiple of 256, height not a multiple of the 32-row strip, one keypoint, 64 keypoints,
    empty batch; bad arguments fail loudly."""
    rng = np.random.default_rng(5)
    for (B, N, h, w) in [(1, 1, 1, 1), (2, 64, 33, 257), (1, 5, 70, 300), (3, 57, 68, 120)]:
        kp = np.stack([rng.uniform(-3, w + 3, (B, N)), rng.uniform(-3, h + 3, (B, N)), (rng.uniform(size=(B, N)) < 0.8)], -1).astype(np.float32)
        got = sncal.loss.create_target(torch.from_numpy(kp).to(cuda), 1.5, (h, w)).cpu().numpy()
        assert np.array_equal(got, synth.create_target(kp, 1.5, (h, w))), (B, N, h, w)
    assert sncal.loss.create_target(torch.zeros((0, 57, 3), device=cuda), 3.0, (8, 8)).shape == (0, 58, 8, 8)
    with pytest.raises(sncal._lib.SncalError, match='N='):
        sncal.loss.create_target(torch.zeros((1, 65, 3), device=cuda), 3.0, (8, 8))
    with pytest.raises(sncal._lib.SncalError, match='sigma'):
        sncal.loss.create_target(torch.zeros((1, 5, 3), device=cuda), 0.0, (8, 8))
