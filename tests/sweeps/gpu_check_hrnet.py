"""Dev helper (GPU box): run the HIP HRNet against goldens / the torch oracle and print error stats."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import sncal_amd  # noqa: E402
from oracle import hrnet_ref as hr  # noqa: E402
from oracle import decode as od  # noqa: E402

dev = torch.device('cuda:0')


def check(name, cfgn, dtype, line=False):
    g = np.load(os.path.join(ROOT, 'tests', 'golden', name + '.npz'))
    cfg = hr.load_config(cfgn)
    sd = hr.seeded_state_dict(cfg, int(g['seed']), float(g['head_gain']))
    x = hr.seeded_input(int(g['batch']), int(g['hw'][0]), int(g['hw'][1]), int(g['seed']) + 1)
    net = sncal_amd.HRNetHeatmap(cfgn, dtype=dtype, device=dev)
    net.load_state_dict(sd)
    t0 = time.time()
    heat, kp = net.forward(x.to(dev), want_heat=True, decode_size=None if line else (540, 960))
    torch.cuda.synchronize()
    t1 = time.time()
    heat = heat.cpu().numpy()
    if 'out' in g:
        ref = g['out']
        err = np.abs(heat - ref)
    else:
        ref = g['out_strided']
        err = np.abs(heat[:, :, ::16, ::16] - ref)
    print(f'{name} [{dtype}] out max|err|={err.max():.3e} mean={err.mean():.3e} ref_absmax={np.abs(ref).max():.3g} '
          f'finite={np.isfinite(heat).all()} t={t1 - t0:.3f}s')
    if not line:
        dec = g['decode']
        k = kp.cpu().numpy()
        same = (k[..., :2] == dec[..., :2]).all(-1)
        gap = g['gap'] / np.maximum(g['maxp'], 1e-30)
        print(f'   keypoint index agreement {same.mean() * 100:.1f}%  (channels with rel gap>1e-3: '
              f'{same[:, :57][gap[:, :57] > 1e-3].mean() * 100:.1f}%)  conf max err {np.abs(k[..., 2] - dec[..., 2]).max():.3e}')
        # decode of our own heat must equal the oracle decode of our heat (fused path consistency)
        assert np.array_equal(k, od.keypoint_decode(heat, (540, 960)))
    return heat


if __name__ == '__main__':
    which = sys.argv[1:] or ['small']
    if 'small' in which:
        for dt in ('fp32', 'bf16'):
            check('hrnet_w18_64x96', 'hrnet_w18', dt)
            check('hrnet_w18_135x240', 'hrnet_w18', dt)
            check('line_w18_64x96', 'line_hrnet_w18', dt, line=True)
    if 'w48' in which:
        for dt in ('fp32', 'bf16'):
            check('hrnet_w48_540x960', 'hrnet_w48', dt)
            check('line_w48_540x960', 'line_hrnet_w48', dt, line=True)
