"""CPU: the functional HRNet restatement against goldens captured from the imported reference."""
import os

import numpy as np
import pytest
import torch

from oracle import decode as od
from oracle import hrnet_ref as hr


@pytest.mark.parametrize('name,cfg,line', [('hrnet_w18_64x96', 'hrnet_w18', False),
                                           ('line_w18_64x96', 'line_hrnet_w18', True)])
def test_small_net_matches_reference_golden(gold_dir, name, cfg, line):
    g = np.load(os.path.join(gold_dir, name + '.npz'))
    c = hr.load_config(cfg)
    sd = hr.seeded_state_dict(c, int(g['seed']), float(g['head_gain']))
    x = hr.seeded_input(int(g['batch']), int(g['hw'][0]), int(g['hw'][1]), int(g['seed']) + 1)
    out, inter = hr.forward(sd, x, c, return_intermediates=True)
    assert np.abs(out.numpy() - g['out']).max() < 1e-5
    assert sum(int(v.numel()) for k, v in sd.items() if 'num_batches' not in k) == int(g['n_params'])
    for k in ('stage2', 'stage3', 'stage4'):
        for bi, t in enumerate(inter[k]):
            assert np.abs(t.numpy()[:, ::4] - g[f'{k}.{bi}']).max() < 1e-4
    if line:
        assert np.array_equal(od.line_decode(out.numpy(), 3.0, 4.0)[..., :2], g['decode'][..., :2])
    else:
        assert np.array_equal(od.keypoint_decode(out.numpy(), (540, 960))[..., :2], g['decode'][..., :2])


def test_conv_enumeration_counts():
    units = hr.enumerate_convs(hr.load_config('hrnet_w48'))
    assert len(units) == 307                                   # 307 Conv2d
    assert sum(1 for u in units if u.bn) == 306
    assert units[-2].cin == units[-2].cout == 784 and units[-1].cout == 58
    lu = hr.enumerate_convs(hr.load_config('line_hrnet_w48'))
    assert lu[-2].cin == 720 and lu[-1].cout == 23


def test_algorithmic_macs_match_survey(gold_dir):
    assert hr.conv_macs(hr.load_config('hrnet_w48'), 540, 960) == 253910384640   # 253.91 GMAC
    g = np.load(os.path.join(gold_dir, 'hrnet_w48_540x960.npz'))
    assert int(g['macs']) == 253910384640
