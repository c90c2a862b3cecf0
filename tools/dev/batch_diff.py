"""Which op first makes a frame's tensors depend on the batch it travels in?  Taps every tensor of the first sub-batch in two forwards
(frames lo..hi alone / inside the full batch) and compares them op by op.   python tools/dev/batch_diff.py [dtype] [lo] [hi] [B]"""
import os, sys, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import sncal_amd
from oracle import hrnet_ref as hr
dtype = sys.argv[1] if len(sys.argv) > 1 else 'fp16x3'
lo, hi, B = (int(a) for a in (sys.argv[2:5] + ['10', '41', '64'][len(sys.argv[2:5]):]))
dev = torch.device('cuda:0')
cfg = hr.load_config('hrnet_w48')
net = sncal_amd.HRNetHeatmap('hrnet_w48', dtype=dtype, device=dev)
net.load_state_dict(hr.seeded_state_dict(cfg, 1, 1.5))
x = torch.rand((67, 3, 540, 960), device=dev, generator=torch.Generator(device=dev).manual_seed(3))[:B].contiguous()

def run(inp):
    net.set_profiling(1)
    net.forward(inp, want_heat=False, decode_size=(540, 960))
    torch.cuda.synchronize()
    ops = net.plan_ops()
    net.set_profiling(0)
    taps = {}
    for op in ops:
        if not op['active'] or op['type'] == 'decode':
            continue
        t = op['out']
        if t is None or t < 0:
            continue
        # the form that is certainly written: the split twin where one is alive, the tensor itself otherwise (a tensor whose twin is
        # alive may have no fp32 form at all, and the mid tensor of a fused block is never written: stale workspace bytes)
        tw = net.plan_tensor(t)['twin']
        tid = tw if tw >= 0 and net.plan_tensor(tw)['alive'] else t
        if op.get('kernel') == 'bblockx3_fused' or not net.plan_tensor(tid)['alive']:
            continue
        taps[(op['idx'], tid)] = net.tap(op['idx'], tid)
    net.forward(inp, want_heat=False, decode_size=(540, 960))
    torch.cuda.synchronize()
    net.clear_taps()
    return ops, {k: v.clone() for k, v in taps.items()}

opsA, A = run(x)
opsB, Bt = run(x[lo:hi].contiguous())
n = 0
for op in opsA:
    for (idx, tid), a in A.items():
        if idx != op['idx'] or (idx, tid) not in Bt:
            continue
        b = Bt[(idx, tid)]
        a = a[lo:hi]
        if a.shape != b.shape:
            print('shape differs', op['name'], a.shape, b.shape); continue
        av, bv = a.reshape(-1).view(torch.int32) if a.element_size() == 4 else a.reshape(-1).view(torch.int16), b.reshape(-1).view(torch.int32) if b.element_size() == 4 else b.reshape(-1).view(torch.int16)
        ne = (av != bv)
        if bool(ne.any()):
            w = torch.nonzero(ne.reshape(a.shape))[:4].tolist()
            fin_a = bool(torch.isfinite(a.float()).all()) if a.is_floating_point() else True
            print(f"op {idx:4d} {op['type']:12s} {op.get('name','')[:40]:40s} kernel {op.get('kernel','')[:36]:36s} tensor {tid} shape {tuple(a.shape)} differing {int(ne.sum())} first {w} finite {fin_a}")
            n += 1
    if n >= 6:
        break
print('done, differing tensors listed:', n)
