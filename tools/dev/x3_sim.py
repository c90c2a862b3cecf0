"""Would a split-arithmetic engine be parity-grade, and with WHICH 16-bit type?  Emulate the conv arithmetic in torch on CPU:
x = xh + xl, w = wh + wl (16-bit each), y = conv(xh,wh) + conv(xh,wl) + conv(xl,wh) accumulated in fp32; everything else fp32.

    bf16x3   hi = bf16(x), lo = bf16(x - hi): 16 significand bits per operand, dropped lo.lo term ~2^-16 of a product
    fp16x3   hi = fp16(x), lo = fp16(x - hi): 22 bits, dropped term ~2^-22 -- same MFMA rate (v_mfma_f32_32x32x16_f16).
             fp16 has 5 exponent bits: `lo` falls into subnormals for |x| < 2^-3 and |hi| overflows above 65504, so the operands are
             scaled by powers of two first (weights: per output channel to max |w| in [2^7, 2^8); activations: one global 2^s);
             `ftz` emulates a matrix unit that flushes fp16 subnormals (measured on gfx950 by tools/dev/f16_mfma_probe.hip: it does not)

usage: python tools/dev/x3_sim.py [frames=3] [H=270] [W=480] -> table on stdout (+ tools/scratch/x3_sim.json)"""
import json, os, sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch, torch.nn.functional as F
import bench, sncal_amd
from oracle import hrnet_ref as hr, decode as od
torch.set_num_threads(8)
sd0 = bench.seeded_weights('hrnet_w48', seed=1)
sd = sncal_amd.synth.peaked_state_dict(sd0, deep=True)
cfg = hr.load_config('hrnet_w48')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 3
H = int(sys.argv[2]) if len(sys.argv) > 2 else 270
W = int(sys.argv[3]) if len(sys.argv) > 3 else 480
frames, expect = sncal_amd.synth.stamped_frames(B, seed=4242, size=(H, W))
x = torch.from_numpy(frames)
orig = F.conv2d
F16_MIN_NORMAL = 2.0 ** -14


def split_bf16(t):
    h = t.to(torch.bfloat16).to(torch.float32)
    l = (t - h).to(torch.bfloat16).to(torch.float32)
    return h, l


def split_f16(t, ftz=False):
    t = t.clamp(-65504.0, 65504.0)
    h = t.to(torch.float16).to(torch.float32)
    l = (t - h).to(torch.float16).to(torch.float32)
    if ftz:
        h = torch.where(h.abs() < F16_MIN_NORMAL, torch.zeros_like(h), h)
        l = torch.where(l.abs() < F16_MIN_NORMAL, torch.zeros_like(l), l)
    return h, l


mode = {'m': 'fp32', 'act_shift': 0, 'w_scale': False, 'ftz': False}
stats = {'max_act': 0.0, 'min_nz_act': 1e30}


def conv(inp, w, b=None, stride=1, padding=0, dilation=1, groups=1):
    m = mode['m']
    if m == 'fp32':
        stats['max_act'] = max(stats['max_act'], float(inp.abs().max()))
        return orig(inp, w, b, stride, padding, dilation, groups)
    if m == 'bf16':
        return orig(inp.to(torch.bfloat16).float(), w.to(torch.bfloat16).float(), b, stride, padding, dilation, groups).to(torch.bfloat16).float()
    if m == 'bf16x3':
        ih, il = split_bf16(inp); wh, wl = split_bf16(w)
        post = None
    else:                               # fp16x3
        sa = 2.0 ** mode['act_shift']
        if mode['w_scale']:             # per output channel: max |w| into [2^7, 2^8)
            amax = w.abs().amax(dim=(1, 2, 3)).clamp_min(1e-30)
            e = torch.floor(torch.log2(amax))
            sw = torch.pow(2.0, 7.0 - e).view(-1, 1, 1, 1)
        else:
            sw = torch.ones((w.shape[0], 1, 1, 1))
        ih, il = split_f16(inp * sa, mode['ftz']); wh, wl = split_f16(w * sw, mode['ftz'])
        post = (1.0 / (sa * sw)).view(1, -1, 1, 1)
    y = orig(ih, wl, None, stride, padding, dilation, groups) + orig(il, wh, None, stride, padding, dilation, groups)
    y = y + orig(ih, wh, None, stride, padding, dilation, groups)
    if post is not None:
        y = y * post
    if b is not None:
        y = y + b.view(1, -1, 1, 1)
    return y


F.conv2d = conv
hr.F.conv2d = conv
runs = [('fp32', {}), ('bf16x3', {}),
        ('fp16x3 raw', dict(m='fp16x3', act_shift=0, w_scale=False, ftz=False)),
        ('fp16x3 raw ftz', dict(m='fp16x3', act_shift=0, w_scale=False, ftz=True)),
        ('fp16x3 act*2^4 w-scaled', dict(m='fp16x3', act_shift=4, w_scale=True, ftz=False)),
        ('fp16x3 act*2^4 w-scaled ftz', dict(m='fp16x3', act_shift=4, w_scale=True, ftz=True)),
        ('bf16', {})]
out = {}
for name, kw in runs:
    mode.update(dict(m=name, act_shift=0, w_scale=False, ftz=False))
    mode.update(kw)
    t = time.time()
    out[name] = hr.forward(sd, x, cfg).numpy()
    print(name, 'forward', round(time.time() - t, 1), 's', flush=True)
print('max |activation| entering a convolution (fp32 run):', stats['max_act'])
ref = out['fp32']
kp_ref = od.keypoint_decode(ref, (540, 960))
usable = kp_ref[..., 2] >= 0.2
table = {}
for name, _ in runs[1:]:
    d = np.abs(out[name] - ref)
    kp = od.keypoint_decode(out[name], (540, 960))
    same = (kp[..., :2] == kp_ref[..., :2]).all(-1)
    dc = kp[..., 2][usable] - kp_ref[..., 2][usable]
    table[name] = dict(dlogp_mean=float(d.mean()), dlogp_max=float(d.max()), moved_usable=int((~same[usable]).sum()), usable=int(usable.sum()),
                       agreement_all_rows=float(same.mean()), dconf_signed_mean=float(dc.mean()), dconf_abs_max=float(np.abs(dc).max()))
    print(f'{name:30s} |dlogp| mean {d.mean():.3e} max {d.max():.3e}  moved {table[name]["moved_usable"]} of {table[name]["usable"]}  all rows {same.mean():.5f}'
          f'  dconf signed mean {dc.mean():+.3e} |max| {np.abs(dc).max():.3e}')
os.makedirs('/root/repo/tools/scratch', exist_ok=True)
json.dump(dict(frames=B, size=[H, W], max_act=stats['max_act'], table=table), open('/root/repo/tools/scratch/x3_sim.json', 'w'), indent=1)
