"""CPU: the oracle's stopping-rule switches (oracle/solve.py STOP; VERDICT r1 task 7).  The build's specification runs every minimiser to
convergence; `opencv_stops()` caps them on the schedule SURVEY 8c attributes to OpenCV 4.7, `iac_failure='reference'` goes on with K = I
where prediction.py:514 ignores the failed factorisation.  On well-conditioned synthetic frames all settings must agree closely -- the
5000-frame sweep (tools/solve_schedule_sweep.py, profiles/r02_solve_schedule_sweep_5000.json) measures how often hard frames differ.
Oracle-vs-oracle: this bounds the unpinned gap, it is not OpenCV parity."""
import contextlib
import io

import numpy as np

from oracle import solve, synth


def _solve(kp):
    with contextlib.redirect_stdout(io.StringIO()):
        return solve.CameraCreatorOracle()(kp, None)


def test_stop_modes_agree_on_well_conditioned_frames():
    saved = dict(solve.STOP)
    try:
        both, close = 0, 0
        for seed in (3, 5, 8, 11, 14, 21):
            kp, _ = synth.synth_keypoints(seed, sigma_px=0.5)
            solve.converged_stops()
            solve.STOP['iac_failure'] = 'drop'
            a = _solve(kp)
            solve.opencv_stops()
            b = _solve(kp)
            solve.converged_stops()
            solve.STOP['iac_failure'] = 'reference'
            c = _solve(kp)
            assert (a is None) == (c is None)                       # no IAC failure on these frames: C == A
            if a is not None and c is not None:
                assert abs(a.rmse - c.rmse) <= 1e-9 * max(1.0, a.rmse)
            if a is not None and b is not None:
                both += 1
                close += abs(a.rmse - b.rmse) <= 1e-2 * a.rmse
        assert both >= 4 and close >= both - 1                      # capped and converged minimisers land in the same basin
    finally:
        solve.STOP.clear()
        solve.STOP.update(saved)


def test_stop_tables():
    saved = dict(solve.STOP)
    try:
        solve.opencv_stops()
        capped = dict(solve.STOP)
        solve.converged_stops()
        # OpenCV: 30 joint iterations, refinement stops on a 1e-5 step / residual; the build: run on until the step is 1e-10
        assert capped['joint_iters'] < solve.STOP['joint_iters'] and capped['pose_eps'] > solve.STOP['pose_eps']
        assert solve.STOP['iac_failure'] in ('drop', 'reference')
    finally:
        solve.STOP.clear()
        solve.STOP.update(saved)
