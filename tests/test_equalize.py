"""CPU: the fp16x3 engine's load-time rebalancing of block-internal channels (sncal_hrnet_equalize, csrc/hrnet.cpp equalize_blocks)
against a numpy restatement of its rule, through the C ABI without a GPU (sncal_hrnet_create / set_conv / equalize / get_conv are host
only).  No reference counterpart: the reference's predict() is fp32 (src/models/hrnet/metamodel.py:127-134); what the step must
preserve is the reference's FUNCTION -- producer row c x 2^-l and consumer column c x 2^l is the same fp32 network bit for bit
(src/models/hrnet/hrnet.py:42-58, 79-99) -- which the test checks on the read-back parameters."""
import numpy as np
import pytest
import torch

import sncal_amd
from oracle import hrnet_ref as hr

BN_EPS = 1e-5


def _fold(sd, units):
    out = []
    for name, bn, cin, cout, k, stride, has_bias in units:
        w = sd[name + '.weight'].double().numpy()
        b = sd[name + '.bias'].double().numpy() if has_bias else np.zeros(cout)
        if bn:
            scale = sd[bn + '.weight'].double().numpy() / np.sqrt(sd[bn + '.running_var'].double().numpy() + BN_EPS)
            shift = sd[bn + '.bias'].double().numpy() + (b - sd[bn + '.running_mean'].double().numpy()) * scale
        else:
            scale, shift = np.ones(cout), b
        out.append([w.astype(np.float32), scale.astype(np.float32), shift.astype(np.float32)])
    return out


def _expected(units, folded, min_log2=4, centre=2):
    """The rule, restated: l_c = round(log2(a_c / m_c) / 2) - centre, moved when |l_c| >= min_log2."""
    moved, exps = 0, {}
    for i in range(len(units) - 1):
        name, bn, nxt = units[i][0], units[i][1], units[i + 1][0]
        stem, leaf = name.rsplit('.', 1)
        nstem, nleaf = nxt.rsplit('.', 1)
        if not bn or stem != nstem or stem == 'model' or (leaf, nleaf) not in (('conv1', 'conv2'), ('conv2', 'conv3')):
            continue
        w1, sc1, sh1 = (a.astype(np.float64) for a in folded[i])
        w2, sc2, _ = (a.astype(np.float64) for a in folded[i + 1])
        m = np.quantile(np.abs(w2).max(axis=(2, 3)) * np.abs(sc2)[:, None], 0.9, axis=0)
        a = np.abs(sh1) + np.sqrt(((w1 * sc1[:, None, None, None]) ** 2).sum(axis=(1, 2, 3)))
        ok = (m > 0) & (a > 0) & np.isfinite(m) & np.isfinite(a)
        lg = np.where(ok, np.rint(0.5 * np.log2(np.where(ok, a, 1.0) / np.where(ok, m, 1.0))) - centre, 0.0)
        lg = np.clip(np.where(np.abs(lg) >= min_log2, lg, 0.0), -60, 60)
        q = np.exp2(lg).astype(np.float32)
        folded[i][1] = folded[i][1] / q
        folded[i][2] = folded[i][2] / q
        folded[i + 1][0] = folded[i + 1][0] * q[None, :, None, None]
        moved += int((lg != 0).sum())
        exps[i] = lg
    return moved, exps


@pytest.mark.parametrize('seed', [0, 1])
def test_library_rebalancing_equals_the_restated_rule_and_keeps_the_function(seed):
    cfg = hr.load_config('hrnet_w18')
    sd = hr.seeded_state_dict(cfg, 9, 4.0)
    net = sncal_amd.HRNetHeatmap('hrnet_w18', dtype='fp16x3', device='cpu')
    units = net.conv_units()
    # an ordinary checkpoint: nothing moves, every parameter is handed on bit for bit
    net.set_convs(sd)
    assert net.equalized == 0
    plain = _fold(sd, units)
    for i in (0, 5, len(units) - 1):
        for got, want in zip(net.folded_conv(i), plain[i]):
            assert np.array_equal(got, want)
    # block-internal scales spread over four decades (same fp32 function): the library brings them back
    sdk = sncal_amd.synth.rescaled_state_dict(sd, units, seed=seed, sigma_log2=3.0, dead_frac=0.02)
    net.set_convs(sdk)
    want = _fold(sdk, units)
    before = [[a.copy() for a in u] for u in want]
    moved, exps = _expected(units, want)
    assert net.equalized == moved and moved > 100
    for i in range(len(units)):
        for got, exp in zip(net.folded_conv(i), want[i]):
            assert np.array_equal(got, exp), units[i][0]
    # the function is preserved exactly: every moved parameter is the original times a power of two, inverse on the consumer column
    for i, lg in exps.items():
        q = np.exp2(lg)
        assert np.array_equal(want[i][1].astype(np.float64) * q, before[i][1].astype(np.float64))
        assert np.array_equal(want[i + 1][0].astype(np.float64), before[i + 1][0].astype(np.float64) * q[None, :, None, None])
    # switched off (sncal_hrnet_set_equalize(net, 0)): parameters untouched
    net.equalize = False
    net.set_convs(sdk)
    assert net.equalized == 0
    assert np.array_equal(net.folded_conv(7)[0], before[7][0])


def test_other_engines_never_rebalance():
    cfg = hr.load_config('hrnet_w18')
    sd = hr.seeded_state_dict(cfg, 9, 4.0)
    net = sncal_amd.HRNetHeatmap('hrnet_w18', dtype='fp32', device='cpu')
    sdk = sncal_amd.synth.rescaled_state_dict(sd, net.conv_units(), seed=0, sigma_log2=3.0, dead_frac=0.02)
    net.set_convs(sdk)
    assert net.equalized == 0
