"""Dev helper: time the HIP HRNet forward (+decode) at BASELINE config C3 and print frames/s."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sncal_amd
from bench import seeded_weights     # the bench's own weight generator: measurement scripts never import oracle/
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dtype = sys.argv[2] if len(sys.argv) > 2 else 'bf16'
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
dev = torch.device('cuda:0')
net = sncal_amd.HRNetHeatmap('hrnet_w48', dtype=dtype, device=dev)
net.load_state_dict(seeded_weights('hrnet_w48', 1))
H, W = int(os.environ.get('DEV_H', '540')), int(os.environ.get('DEV_W', '960'))
x = torch.rand((B, 3, H, W), device=dev)
if dtype == 'fp8':
    net.calibrate_fp8(x[:8])
for _ in range(1):
    net.forward(x, want_heat=False, decode_size=(H, W))
torch.cuda.synchronize()
net.set_profiling(os.environ.get('DEV_NOPROF', '0') != '1')
t0 = time.time()
for _ in range(steps):
    net.forward(x, want_heat=False, decode_size=(H, W))
torch.cuda.synchronize()
dt = (time.time() - t0) / steps
print(f'B={B} {dtype} subbatch={os.environ.get("SNCAL_SUBBATCH", "64")}: {dt*1e3:.1f} ms/step, {B/dt:.1f} frames/s, {B/dt*507.82e9/1e12:.1f} TFLOP/s (reference-formulation flops)')

prof = sorted(net.get_profile(), key=lambda p: -p['ms'])
tot = sum(p['ms'] for p in prof)
for p in prof[:int(os.environ.get("DEV_TOP", "14"))]:
    tf = p['flops'] / (p['ms'] * 1e-3) / 1e12 if p['ms'] else 0
    gbs = p['bytes'] / (p['ms'] * 1e-3) / 1e9 if p['ms'] else 0
    print(f"  {p['ms']/tot*100:5.1f}%  {p['ms']/steps:8.2f} ms/step  n={p['launches']//steps:4d}  {tf:7.1f} TF  {gbs:7.0f} GB/s  {p['kernel']}")
