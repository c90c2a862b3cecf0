"""Round 6: frames on which the HIP solve and the numpy oracle disagree (tests/sweeps/gpu_check_solve_all.py 5000): what each HIP path
returns for them.  python tools/dev/solve_diff_probe.py 1063 1943 1626 2680 3136"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import sncal_amd
from oracle import solve, synth
frames = [int(a) for a in sys.argv[1:]] or [1063, 1943]
kps = np.stack([synth.synth_keypoints(s, sigma_px=1.0)[0] for s in frames])
KW = dict(conf_thresh=0.5, conf_threshs=[0.5, 0.35, 0.2], max_rmse=55.0, max_rmse_rel=5.0, min_points=5, min_focal_length=10.0,
          min_points_per_plane=6, min_points_for_refinement=6, reliable_thresh=57)
for alg in ('voter', 'iterative_voter', 'original_voter'):
    for sched in ('opencv', 'converged'):
        cc = sncal_amd.CameraCreator(sncal_amd.PITCH_POINTS, algorithm=alg, lm_schedule=sched, **KW)
        recs = cc.records(cc.solve_device(torch.from_numpy(kps).cuda()))
        oc = solve.CameraCreatorOracle(algorithm=alg)
        solve.converged_stops() if sched == 'converged' else solve.opencv_stops()
        for f, k, r in zip(frames, kps, recs):
            o = oc(k, None)
            print(alg, sched, 'frame', f, 'HIP', (r.status, round(r.rmse, 4), round(r.fx, 2), r.n_points), 'oracle', None if o is None else (o.tag, round(o.rmse, 4), round(o.xfocal_length, 2)))
