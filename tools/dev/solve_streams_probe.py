"""Do camera solves on different streams overlap?  N noisy keypoint frames (tools/noisy_pipeline.py) as batches of 64 on P streams in
turn, no network running: total wall time per P.  GPU box:  python tools/dev/solve_streams_probe.py [N=1024] [P ...]"""
import os, sys, time
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tools'))
import numpy as np, torch
import sncal_amd, bench
from noisy_pipeline import noisy_keypoints
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
Ps = [int(a) for a in sys.argv[2:]] or [1, 2, 4, 8]
kp = torch.from_numpy(noisy_keypoints(N)).cuda()
cc = sncal_amd.CameraCreator(sncal_amd.PITCH_POINTS, **bench.SOLVER_KW)
cc.solve_device(kp[:64].contiguous()); torch.cuda.synchronize()
for P in Ps:
    streams = [torch.cuda.Stream() for _ in range(P)]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    outs = []
    for b in range(N // 64):
        with torch.cuda.stream(streams[b % P]):
            outs.append(cc.solve_device(kp[b * 64:(b + 1) * 64].contiguous()))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f'P={P}: {dt * 1e3:.1f} ms for {N // 64} batches = {dt * 1e3 / (N // 64):.1f} ms per batch', flush=True)
