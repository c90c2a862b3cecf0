"""A/B: iterative_voter in one kernel (SNCAL_SOLVE_SPLIT=0) vs original_voter kernel + 4-wave voter kernel: records must be
byte-identical; prints both timings.  Run once per setting (the env var is read once per process)."""
import os, sys, hashlib, numpy as np, torch
sys.path.insert(0, '.')
import sncal_amd
dev = torch.device('cuda:0')
cc = sncal_amd.CameraCreator(sncal_amd.PITCH_POINTS, conf_thresh=0.5, conf_threshs=[0.5, 0.35, 0.2], algorithm='iterative_voter', lines_file=None, max_rmse=55.0, max_rmse_rel=5.0, min_points=5, min_focal_length=10.0, min_points_per_plane=6, min_points_for_refinement=6, reliable_thresh=57)
h = hashlib.sha256()
tags = np.zeros(8, int)
for seed in range(8):
    kp = sncal_amd.synth.synthetic_keypoints(64, seed=300 + seed, min_visible=4 + seed)
    if seed >= 4:                       # harder frames: more noise / lower confidences so that the voter passes are exercised
        rng = np.random.default_rng(seed)
        kp[..., 2] *= rng.uniform(0.3, 1.0, kp[..., 2].shape).astype(np.float32)
        kp[..., :2] += rng.normal(0, 2.0, kp[..., :2].shape).astype(np.float32)
    rec = cc.solve_device(torch.from_numpy(kp).to(dev))
    raw = rec.cpu().numpy().tobytes()
    h.update(raw)
    for r in cc.records(rec):
        tags[r.status] += 1
kp = torch.from_numpy(sncal_amd.synth.synthetic_keypoints(64, seed=77)).to(dev)
rec = cc.solve_device(kp); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): cc.solve_device(kp, out=rec)
e1.record(); torch.cuda.synchronize()
print('split', os.environ.get('SNCAL_SOLVE_SPLIT', '1'), 'sha', h.hexdigest()[:16], 'tags', tags.tolist(), 'bench batch ms %.3f' % (e0.elapsed_time(e1) / 5))
