"""Would a split-bf16 (bf16x3) engine be parity-grade?  Emulate its conv arithmetic in torch on CPU: x = xh + xl, w = wh + wl (bf16 each),
y = conv(xh,wh) + conv(xh,wl) + conv(xl,wh) accumulated in fp32; everything else fp32.  Compare with the plain fp32 forward and with bf16."""
import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch, torch.nn.functional as F
import bench, sncal_amd
from oracle import hrnet_ref as hr, decode as od
torch.set_num_threads(8)
sd0 = bench.seeded_weights('hrnet_w48', seed=1)
sd = sncal_amd.synth.peaked_state_dict(sd0, deep=True)
cfg = hr.load_config('hrnet_w48')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 3
frames, expect = sncal_amd.synth.stamped_frames(B, seed=4242, size=(270, 480))
x = torch.from_numpy(frames)
orig = F.conv2d
def split(t):
    h = t.to(torch.bfloat16).to(torch.float32)
    l = (t - h).to(torch.bfloat16).to(torch.float32)
    return h, l
mode = {'m': 'fp32'}
def conv(inp, w, b=None, stride=1, padding=0, dilation=1, groups=1):
    if mode['m'] == 'fp32':
        return orig(inp, w, b, stride, padding, dilation, groups)
    if mode['m'] == 'bf16':
        return orig(inp.to(torch.bfloat16).float(), w.to(torch.bfloat16).float(), b, stride, padding, dilation, groups).to(torch.bfloat16).float()
    ih, il = split(inp); wh, wl = split(w)
    y = orig(ih, wh, None, stride, padding, dilation, groups) + orig(ih, wl, None, stride, padding, dilation, groups) + orig(il, wh, None, stride, padding, dilation, groups)
    if mode['m'] == 'x4':
        y = y + orig(il, wl, None, stride, padding, dilation, groups)
    if b is not None:
        y = y + b.view(1, -1, 1, 1)
    return y
F.conv2d = conv
hr.F.conv2d = conv
out = {}
for m in ('fp32', 'x3', 'bf16'):
    mode['m'] = m
    t = time.time()
    out[m] = hr.forward(sd, x, cfg).numpy()
    print(m, 'forward', round(time.time() - t, 1), 's')
# BN is applied separately in the oracle (not folded), fine: same in all modes
ref = out['fp32']
kp_ref = od.keypoint_decode(ref, (540, 960))
usable = kp_ref[..., 2] >= 0.2
for m in ('x3', 'bf16'):
    d = np.abs(out[m] - ref)
    kp = od.keypoint_decode(out[m], (540, 960))
    same = (kp[..., :2] == kp_ref[..., :2]).all(-1)
    print(m, '|dlogp| mean', d.mean(), 'max', d.max(), 'index agreement usable', same[usable].mean(), f'({int((~same[usable]).sum())} of {int(usable.sum())} moved)', 'all rows', same.mean())
