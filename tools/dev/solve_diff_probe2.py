"""Frame 1063 of the solve sweep reduced to its H-consistent subset (the voter's camera_acc candidate): HIP vs oracle per algorithm / schedule."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import sncal_amd
from oracle import solve, synth
kp, _ = synth.synth_keypoints(1063, sigma_px=1.0)
kp2 = kp.copy(); kp2[44, 2] = 0.0                      # without the gross outlier
kp3 = kp2.copy(); kp3[0, 2] = 0.0                      # ... and without the crossbar point: the goal plane drops below 6 points
KW = dict(conf_thresh=0.5, conf_threshs=[0.5, 0.35, 0.2], max_rmse=55.0, max_rmse_rel=5.0, min_points=5, min_focal_length=10.0,
          min_points_per_plane=6, min_points_for_refinement=6, reliable_thresh=57)
for name, k in (('full', kp), ('no outlier', kp2), ('no outlier, no crossbar', kp3)):
    for alg in ('voter', 'original_voter', 'opencv_calibration_multiplane', 'opencv_calibration'):
        for sched in ('opencv', 'converged'):
            cc = sncal_amd.CameraCreator(sncal_amd.PITCH_POINTS, algorithm=alg, lm_schedule=sched, **KW)
            r = cc.records(cc.solve_device(torch.from_numpy(k[None]).cuda()))[0]
            solve.converged_stops() if sched == 'converged' else solve.opencv_stops()
            o = solve.CameraCreatorOracle(algorithm=alg)(k, None)
            print(f'{name:24s} {alg:30s} {sched:9s} HIP', (r.status, round(r.rmse, 4), round(r.fx, 2)), 'oracle', None if o is None else (getattr(o, 'tag', ''), round(o.rmse, 4) if hasattr(o, 'rmse') and o.rmse is not None else None, round(o.xfocal_length, 2)))
