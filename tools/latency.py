"""Forward + fused decode time of the HRNet-W48 960x540 engines by batch size (network only, no solve): python tools/latency.py"""
import os, sys, time, json
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sncal_amd
from bench import seeded_weights
dev = torch.device('cuda:0')
sd = seeded_weights('hrnet_w48', 1)
rows = {}
for dtype in ('fp32', 'fp16x3', 'bf16'):
    net = sncal_amd.HRNetHeatmap('hrnet_w48', dtype=dtype, device=dev)
    net.load_state_dict(sd)
    for B in (1, 2, 4, 8, 16, 32, 64):
        x = torch.rand((B, 3, 540, 960), device=dev)
        for _ in range(2):
            net.forward(x, want_heat=False, decode_size=(540, 960))
        torch.cuda.synchronize()
        n = max(3, 64 // B)
        t0 = time.perf_counter()
        for _ in range(n):
            net.forward(x, want_heat=False, decode_size=(540, 960))
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / n * 1e3
        rows.setdefault(dtype, {})[B] = round(ms, 2)
        print(f'{dtype:7s} B={B:3d}: {ms:8.2f} ms per forward, {B / ms * 1e3:8.1f} frames/s', flush=True)
os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
json.dump(rows, open(os.path.join(ROOT, 'gpurun_out', 'latency_by_batch.json'), 'w'), indent=1)
