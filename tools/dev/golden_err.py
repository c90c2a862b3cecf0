"""Distance of each engine from the reference-captured W48 goldens (tests/golden): max / mean |d log p| (keypoint net), max |d p| (line net),
confidence delta.   python tools/dev/golden_err.py"""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import sncal_amd
import test_hrnet_gpu as T
dev = torch.device('cuda:0')
gd = os.path.join(ROOT, 'tests', 'golden')
x3 = sncal_amd._lib.lib().sncal_x3_name().decode()
for dtype in ('fp32', x3, 'bf16'):
    g, heat, kp = T._run(sncal_amd, dev, gd, 'hrnet_w48_540x960', 'hrnet_w48', dtype)
    ref = g['out'] if 'out' in g else g['out_strided']
    got = heat if 'out' in g else heat[:, :, ::16, ::16]
    fin = np.isfinite(ref)
    d = np.abs(got - ref)[fin]
    print(f'keypoint net {dtype:7s} |dlogp| max {d.max():.3e} mean {d.mean():.3e}  conf max {np.abs(kp[..., 2] - g["decode"][..., 2]).max():.3e}  indices identical {np.array_equal(kp[..., :2], g["decode"][..., :2])}')
    if dtype != 'bf16':
        g, heat, _ = T._run(sncal_amd, dev, gd, 'line_w48_540x960', 'line_hrnet_w48', dtype, line=True)
        print(f'line net     {dtype:7s} |dp| max {T._err(g, heat):.3e}')
