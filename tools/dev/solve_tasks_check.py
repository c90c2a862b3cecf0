"""The one-wavefront-per-camera voter (sncal_calibrate's default since round 4) against the four-wave voter_kernel it replaces
(SNCAL_SOLVE_TASKS=0): records byte for byte and time per batch of 64, on N noisy synthetic frames (the library's default refine
criterion AND the bench's 200-iteration cap) and on the bench's own keypoints if gpurun_out/bench_kp.npy exists.  GPU box:
    python tools/dev/solve_tasks_check.py run out.npz [N]      (once per setting of SNCAL_SOLVE_TASKS)
    python tools/dev/solve_tasks_check.py cmp a.npz b.npz"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

if sys.argv[1] == 'cmp':
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    for k in a.files:
        if k.startswith('ms_'):
            print(f'{k}: {a[k].mean():.2f} / {b[k].mean():.2f} ms per batch of 64 (mean), max {a[k].max():.2f} / {b[k].max():.2f}')
        else:
            same = np.array_equal(a[k], b[k])
            rows = int((a[k] != b[k]).any(axis=1).sum()) if not same else 0
            print(f'{k}: {a[k].shape[0]} records, identical bytes: {same}' + ('' if same else f' ({rows} records differ)'))
    sys.exit(0)

import torch
import sncal_amd
from oracle import synth
out = sys.argv[2]
N = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
kps = np.stack([synth.synth_keypoints(s, sigma_px=(0.5, 1.0, 2.0, 4.0)[s % 4])[0] for s in range(N)]).astype(np.float32)
KW = dict(conf_thresh=0.5, conf_threshs=[0.5, 0.35, 0.2], max_rmse=55.0, max_rmse_rel=5.0, min_points=5,
          min_focal_length=10.0, min_points_per_plane=6, min_points_for_refinement=6, reliable_thresh=57)
res = {}
sets = {'synth': kps}
bk = os.path.join(ROOT, 'gpurun_out', 'bench_kp.npy')
if os.path.exists(bk):
    sets['bench'] = np.load(bk)
for cap in (20000, 200):
    cc = sncal_amd.CameraCreator(sncal_amd.PITCH_POINTS, algorithm='iterative_voter', refine_max_iters=cap, **KW)
    for name, k in sets.items():
        d = torch.from_numpy(k).cuda()
        cc.solve_device(d[:64])
        recs, ts = [], []
        for b in range(0, len(k), 64):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            r = cc.solve_device(d[b:b + 64].contiguous())
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
            recs.append(r.cpu().numpy())
        res[f'rec_{name}_cap{cap}'] = np.concatenate(recs)
        res[f'ms_{name}_cap{cap}'] = np.array(ts)
        print(name, cap, f'{np.mean(ts):.2f} ms per batch, max {np.max(ts):.2f}', flush=True)
np.savez(out, **res)
