"""Does replaying one forward as a hipGraph shorten small-batch latency?  (run on the GPU box)"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import sncal_amd
from bench import seeded_weights
dev = torch.device('cuda:0')
sd = seeded_weights('hrnet_w48', 1)
for dtype in ('fp16x3', 'bf16'):
    net = sncal_amd.HRNetHeatmap('hrnet_w48', dtype=dtype, device=dev)
    net.load_state_dict(sd)
    for B in (1, 8, 64):
        x = torch.rand((B, 3, 540, 960), device=dev)
        for _ in range(3):
            k0 = net.forward(x, want_heat=False, decode_size=(540, 960))
        torch.cuda.synchronize()
        n = max(5, 128 // B)
        t0 = time.perf_counter()
        for _ in range(n):
            net.forward(x, want_heat=False, decode_size=(540, 960))
        torch.cuda.synchronize()
        eager = (time.perf_counter() - t0) / n * 1e3
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            net.forward(x, want_heat=False, decode_size=(540, 960))
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        try:
            with torch.cuda.graph(g):
                k1 = net.forward(x, want_heat=False, decode_size=(540, 960))
            for _ in range(3):
                g.replay()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                g.replay()
            torch.cuda.synchronize()
            graph = (time.perf_counter() - t0) / n * 1e3
            same = bool(torch.equal(k0, k1)) if isinstance(k0, torch.Tensor) else None
            print(f'{dtype:7s} B={B:3d}: eager {eager:7.2f} ms, graph replay {graph:7.2f} ms, same keypoints {same}', flush=True)
        except Exception as e:
            print(f'{dtype:7s} B={B:3d}: eager {eager:7.2f} ms, capture failed: {e!r}', flush=True)
