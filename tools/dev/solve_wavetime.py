"""Solve the bench's keypoints (tools/scratch/bench_kp.npy) and N noisy frames, one synchronous batch at a time (for a PMC pass)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tools'))
import numpy as np, torch
import sncal_amd, bench
from noisy_pipeline import noisy_keypoints
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
cc = sncal_amd.CameraCreator(sncal_amd.PITCH_POINTS, **bench.SOLVER_KW)
sets = {'noisy': noisy_keypoints(N)}
bk = os.path.join(ROOT, 'tools', 'scratch', 'bench_kp.npy')
if os.path.exists(bk):
    sets['bench'] = np.load(bk)
for name, k in sets.items():
    d = torch.from_numpy(k).cuda()
    for b in range(0, len(k), 64):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        cc.solve_device(d[b:b + 64].contiguous())
        torch.cuda.synchronize()
        print(name, b // 64, f'{(time.perf_counter() - t0) * 1e3:.1f} ms', flush=True)
