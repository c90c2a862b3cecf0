"""Frame 1063 without its outlier, opencv_calibration_multiplane: HIP's calibrate_planes internals (debug build: tools/ab_build.py dbg
-DSNCAL_SOLVE_DEBUG=1, SNCAL_LIB_PATH=tools/ab/libsncal_dbg.so) next to the oracle's."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from oracle import solve, synth
kp, _ = synth.synth_keypoints(1063, sigma_px=1.0)
kp[44, 2] = 0.0
sched = sys.argv[1] if len(sys.argv) > 1 else 'opencv'
if len(sys.argv) > 2 and sys.argv[2] == 'hip':
    import torch, sncal_amd
    cc = sncal_amd.CameraCreator(sncal_amd.PITCH_POINTS, algorithm='opencv_calibration_multiplane', conf_thresh=0.5, min_points=5, min_points_per_plane=6,
                                 min_points_for_refinement=6, min_focal_length=10.0, reliable_thresh=57, lm_schedule=sched)
    r = cc.records(cc.solve_device(torch.from_numpy(kp[None]).cuda()))[0]
    torch.cuda.synchronize()
    print('HIP result', r.status, r.rmse, r.fx)
    sys.exit(0)
solve.converged_stops() if sched == 'converged' else solve.opencv_stops()
oc = solve.CameraCreatorOracle(algorithm='opencv_calibration_multiplane')
ids = [i for i in range(57) if kp[i, 2] > 0.5]
uv = np.array([[float(kp[i, 0]), float(kp[i, 1])] for i in ids])
uv32 = uv.astype(np.float32).astype(np.float64)
views, weights = solve._views_from(ids, uv32, 6, duplicate=False)
print('views', [(len(v[0]), w) for v, w in zip(views, weights)])
cx, cy = 479.5, 269.5
for vi, (Xp, u) in enumerate(views):
    H = solve._homography_plain(Xp[:, :2], u)
    print('[orc] view', vi, 'H', ' '.join('%.10g' % x for x in H.ravel()))
orig = solve.pose_from_homography
def spy(H, fx, fy, cx_, cy_):
    out = orig(H, fx, fy, cx_, cy_)
    print('[orc] f_init %.10g' % fx, 'pose0 R', None if out is None else ' '.join('%.8g' % x for x in out[0].ravel()), 't', None if out is None else out[1])
    return out
solve.pose_from_homography = spy
orig_refit = solve._refit
def spy2(R, t, K4, X, uv_):
    out = orig_refit(R, t, K4, X, uv_)
    print('[orc] refit R', ' '.join('%.8g' % x for x in out[0].ravel()), 't', out[1])
    return out
solve._refit = spy2
res = solve.calibrate_planes(views, weights, (960, 540))
print('[orc] result f', None if res is None else res[0])
