"""Sweep (GPU box; not collected by pytest): the HIP camera solve vs the numpy oracle on N synthetic frames for ALL five algorithms,
the oracle on a process pool.  Writes gpurun_out/solve_sweep_<N>.json.   python tests/sweeps/gpu_check_solve_all.py 1000 [opencv|converged]"""
import json, os, sys, time
import multiprocessing as mp
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
ALGS = ['iterative_voter', 'original_voter', 'voter', 'opencv_calibration', 'opencv_calibration_multiplane']


def _oracle(args):
    alg, k, sched = args
    from oracle import solve
    solve.converged_stops() if sched == 'converged' else solve.opencv_stops()
    o = solve.CameraCreatorOracle(algorithm=alg)(k, None)
    return None if o is None else (float(o.rmse), float(o.xfocal_length), [float(v) for v in o.position])


if __name__ == '__main__':
    import torch
    import sncal_amd
    from oracle import synth
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    SCHED = sys.argv[2] if len(sys.argv) > 2 else 'opencv'           # 'opencv' (default: OpenCV's iteration-capped schedules) or 'converged'
    kps = np.stack([synth.synth_keypoints(s, sigma_px=1.0)[0] for s in range(N)])
    out = {'frames': N, 'lm_schedule': SCHED, 'noise_px': 1.0, 'tolerance': '1e-4 relative reprojection error; None-ness identical', 'algorithms': {}}
    ctx = mp.get_context('spawn')
    with ctx.Pool(min(16, os.cpu_count() or 1)) as pool:
        for alg in ALGS:
            cc = sncal_amd.CameraCreator(sncal_amd.PITCH_POINTS, conf_thresh=0.5, conf_threshs=[0.5, 0.35, 0.2], algorithm=alg, max_rmse=55.0,
                                         max_rmse_rel=5.0, min_points=5, min_focal_length=10.0, min_points_per_plane=6,
                                         min_points_for_refinement=6, reliable_thresh=57, lm_schedule=SCHED)
            t0 = time.time()
            cams = cc.solve_batch(kps)
            torch.cuda.synchronize()
            t_hip = time.time() - t0
            t0 = time.time()
            ref = pool.map(_oracle, [(alg, k, SCHED) for k in kps], chunksize=4)
            t_or = time.time() - t0
            none_mismatch, both, within, worst, bad = 0, 0, 0, 0.0, []
            for i, (c, o) in enumerate(zip(cams, ref)):
                if (c is None) != (o is None):
                    none_mismatch += 1
                    bad.append({'frame': i, 'oracle': o, 'hip': None if c is None else float(c.rmse)})
                    continue
                if o is None:
                    continue
                both += 1
                rel = abs(o[0] - c.rmse) / max(o[0], 1e-12)
                worst = max(worst, rel)
                if rel <= 1e-4:
                    within += 1
                else:
                    bad.append({'frame': i, 'oracle_rmse': o[0], 'hip_rmse': float(c.rmse), 'oracle_f': o[1], 'hip_f': float(c.xfocal_length)})
            out['algorithms'][alg] = {'cameras_both': both, 'within_1e-4': within, 'none_ness_mismatches': none_mismatch, 'worst_rel': worst,
                                      'hip_s': round(t_hip, 2), 'oracle_pool_s': round(t_or, 1), 'differing': bad[:8]}
            print(alg, out['algorithms'][alg], flush=True)
    os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, 'gpurun_out', f'solve_sweep_{N}' + ('' if SCHED == 'opencv' else '_' + SCHED) + '.json'), 'w'), indent=1)
