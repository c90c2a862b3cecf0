"""Dev helper (GPU box): HIP camera solve vs the numpy oracle on synthetic keypoints."""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import sncal_amd
from oracle import solve, synth

N = int(sys.argv[1]) if len(sys.argv) > 1 else 48
alg = sys.argv[2] if len(sys.argv) > 2 else 'iterative_voter'
kps = np.stack([synth.synth_keypoints(s, sigma_px=1.0)[0] for s in range(N)])
cc = sncal_amd.CameraCreator(sncal_amd.PITCH_POINTS, conf_thresh=0.5, conf_threshs=[0.5, 0.35, 0.2], algorithm=alg,
                             max_rmse=55.0, max_rmse_rel=5.0, min_points=5, min_focal_length=10.0,
                             min_points_per_plane=6, min_points_for_refinement=6, reliable_thresh=57)
t0 = time.time(); cams = cc.solve_batch(kps); torch.cuda.synchronize(); t1 = time.time()
t0 = time.time(); cams = cc.solve_batch(kps); torch.cuda.synchronize(); t1 = time.time()
print(f'HIP solve of {N} frames: {1e3 * (t1 - t0):.1f} ms')
oc = solve.CameraCreatorOracle(algorithm=alg)
bad = 0
for i in range(N):
    o = oc(kps[i], None)
    c = cams[i]
    if (o is None) != (c is None):
        print(i, 'MISMATCH none-ness: oracle', None if o is None else (o.tag, o.rmse), 'hip', None if c is None else (c.source, c.rmse)); bad += 1; continue
    if o is None:
        continue
    rel = abs(o.rmse - c.rmse) / max(o.rmse, 1e-12)
    flag = '' if rel < 1e-4 else '   <<<<<'
    bad += rel >= 1e-4
    print(f'{i:3d} {o.tag:14s} {c.source:18s} rmse oracle {o.rmse:.6f} hip {c.rmse:.6f} rel {rel:.2e} f {o.xfocal_length:.3f}/{c.xfocal_length:.3f} dpos {np.linalg.norm(o.position - c.position):.2e}{flag}')
print('mismatches:', bad, 'of', N)
