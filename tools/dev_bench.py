"""Dev helper: time the HIP HRNet forward (+decode) at BASELINE config C3 and print frames/s."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sncal_amd
from oracle import hrnet_ref as hr
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dtype = sys.argv[2] if len(sys.argv) > 2 else 'bf16'
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
dev = torch.device('cuda:0')
cfg = hr.load_config('hrnet_w48')
net = sncal_amd.HRNetHeatmap('hrnet_w48', dtype=dtype, device=dev)
net.load_state_dict(hr.seeded_state_dict(cfg, 1, 1.5))
x = torch.rand((B, 3, 540, 960), device=dev)
for _ in range(1):
    net.forward(x, want_heat=False, decode_size=(540, 960))
torch.cuda.synchronize()
t0 = time.time()
for _ in range(steps):
    net.forward(x, want_heat=False, decode_size=(540, 960))
torch.cuda.synchronize()
dt = (time.time() - t0) / steps
print(f'B={B} {dtype} subbatch={os.environ.get("SNCAL_SUBBATCH", "8")}: {dt*1e3:.1f} ms/step, {B/dt:.1f} frames/s, {B/dt*507.82e9/1e12:.1f} TFLOP/s (reference-formulation flops)')
