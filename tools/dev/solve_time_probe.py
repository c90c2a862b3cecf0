"""Time of the batched solve (iterative_voter, the library's default refine criterion) on N synthetic frames, per batch of 64 (GPU box)."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import sncal_amd
from oracle import synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
kps = np.stack([synth.synth_keypoints(s, sigma_px=(0.5, 1.0, 2.0)[s % 3])[0] for s in range(N)])
KW = dict(conf_thresh=0.5, conf_threshs=[0.5, 0.35, 0.2], max_rmse=55.0, max_rmse_rel=5.0, min_points=5,
          min_focal_length=10.0, min_points_per_plane=6, min_points_for_refinement=6, reliable_thresh=57)
cc = sncal_amd.CameraCreator(sncal_amd.PITCH_POINTS, algorithm='iterative_voter', **KW)
cc.solve_batch(kps[:64])
ts = []
none = 0
for b in range(0, N, 64):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    cams = cc.solve_batch(kps[b:b + 64])
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) * 1e3)
    none += sum(c is None for c in cams)
ts = np.array(ts)
print(f'{N} frames: per batch of 64 mean {ts.mean():.1f} ms, median {np.median(ts):.1f}, max {ts.max():.1f}; frames without a camera {none}')
