"""OPTIONAL: the oracle's restatements of the OpenCV routines against OpenCV itself.  Skipped wherever cv2 is not
importable -- which includes the build container and the GPU box: OpenCV parity of the solve stays UNPINNED until this file has run
somewhere with opencv-python==4.7.0.72 and tests/golden/solve_cv2.npz (tools/make_solve_cv2_fixture.py) is committed.

Per primitive, on the seeded frames of the fixture generator, with the reference's own flags and criteria
(prediction.py:398-408, 614-623; baseline/camera.py:100-102, 112-118; src/datatools/ellipse.py:497):
    homography_ransac   vs cv2.findHomography(RANSAC, 10)       transfer of the ground points within 1 px (RANSAC draws differ)
    calibrate_planes    vs cv2.calibrateCamera                  focal length to 1e-4 relative, pose of view 0
    lm_solver_pose      vs cv2.solvePnPRefineLM                 from OpenCV's own starting pose: rvec / tvec to 1e-6
    pnp_ransac + refine vs cv2.solvePnPRansac + RefineLM        reprojection error to 1e-4 relative (the north-star tolerance)
The expectations were written WITHOUT a cv2 at hand; a failure here is information about the restatement, not a broken build.

Every check exists twice: unmarked (the CPU run, `-m "not gpu"`) and `gpu`-marked (the driver's `-m gpu` run on the GPU box), so that
whichever box first carries a cv2 runs them.  When the committed fixture is missing, the first run with a cv2 WRITES it --
tests/golden/solve_cv2.npz and a copy under gpurun_out/ (which travels back from a GPU box) -- so that one such run is enough to pin
the oracle from then on."""
import os

import numpy as np
import pytest

cv2 = pytest.importorskip('cv2')

from oracle import solve  # noqa: E402
import importlib.util  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_spec = importlib.util.spec_from_file_location('make_solve_cv2_fixture', os.path.join(ROOT, 'tools', 'make_solve_cv2_fixture.py'))
gen = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(gen)


_ROWS = None


def _rows():
    """cv2's outputs: the committed fixture when present (pinned OpenCV), else the cv2 that is importable now -- whose outputs are
    then saved as the fixture (and beside the GPU box's logs), with the version they came from."""
    global _ROWS
    if _ROWS is not None:
        return _ROWS
    path = os.path.join(ROOT, 'tests', 'golden', 'solve_cv2.npz')
    if os.path.exists(path):
        g = np.load(path)
        _ROWS = [{k: g[k][i] for k in ('kp', 'H', 'cal1', 'calq', 'pnp', 'refine')} | {'seed': int(s)} for i, s in enumerate(g['seeds'])]
        return _ROWS
    rows = [gen.run_cv2(cv2, s) for s in gen.SEEDS]
    payload = dict(seeds=np.array(gen.SEEDS), cv2_version=np.array(cv2.__version__), **{k: np.stack([r[k] for r in rows]) for k in rows[0]})
    for dst in (path, os.path.join(ROOT, 'gpurun_out', 'solve_cv2.npz')):
        try:
            os.makedirs(os.path.dirname(dst), exist_ok=True)
            np.savez_compressed(dst, **payload)
        except OSError:
            pass
    _ROWS = [r | {'seed': s} for r, s in zip(rows, gen.SEEDS)]
    return _ROWS


def _rmse(rvec, t, K4, X, uv):
    p, _ = solve.project(solve.exp_so3(rvec), t, K4, X)
    return float(np.sqrt(((p - uv) ** 2).sum(1)).mean())


def _refine_pose_follows_solvePnPRefineLM():
    n = 0
    for row in _rows():
        if not np.isfinite(row['refine']).all():
            continue
        _, ids, uv, _, _, _ = gen.frame_inputs(row['seed'])
        f, cx, cy = row['calq'][1:4]
        K4 = (f, f, cx, cy)
        R, t = solve.lm_solver_pose(solve.exp_so3(row['pnp'][:3]), row['pnp'][3:], K4, solve.P64[ids], uv, 20000, 1e-5)
        assert np.allclose(solve.log_so3(R), row['refine'][:3], atol=1e-6) and np.allclose(t, row['refine'][3:], atol=1e-5), row['seed']
        n += 1
    assert n > 0


def _calibrate_planes_follows_calibrateCamera():
    n = 0
    for row in _rows():
        for key, dup in (('cal1', False), ('calq', True)):
            if not np.isfinite(row[key]).all():
                continue
            _, ids, uv, v1, vq, wq = gen.frame_inputs(row['seed'])
            res = solve.calibrate_planes(vq if dup else v1, wq if dup else [1] * len(v1), (960, 540))
            assert res is not None, (row['seed'], key)
            f, cx, cy, R0, t0 = res
            assert abs(f - row[key][1]) <= 1e-4 * row[key][1], (row['seed'], key, f, row[key][1])
            assert (cx, cy) == (row[key][2], row[key][3])                     # Q3: ((w-1)/2, (h-1)/2)
            assert np.allclose(solve.log_so3(R0), row[key][4:7], atol=1e-4) and np.allclose(t0, row[key][7:10], rtol=1e-4, atol=1e-3)
            n += 1
    assert n > 0


def _homography_ransac_agrees_with_findHomography_on_the_inliers():
    n = 0
    for row in _rows():
        if not np.isfinite(row['H']).all():
            continue
        _, ids, uv, _, _, _ = gen.frame_inputs(row['seed'])
        g = [k for k, i in enumerate(ids) if i in gen.GROUND]
        src = solve.P32[[ids[k] for k in g], :2]
        uv32 = uv.astype(np.float32).astype(np.float64)
        H = solve.homography_ransac(src, uv32[g], 10.0)
        assert H is not None
        a, b = solve._apply_h(H, src), solve._apply_h(row['H'], src)
        inl = np.sqrt(((b - uv32[g]) ** 2).sum(1)) <= 10.0
        assert np.sqrt(((a - b) ** 2).sum(1))[inl].max() <= 1.0, row['seed']
        n += 1
    assert n > 0


def _pnp_and_refine_reach_the_same_reprojection_error():
    n = 0
    for row in _rows():
        if not np.isfinite(row['refine']).all():
            continue
        _, ids, uv, _, _, _ = gen.frame_inputs(row['seed'])
        f, cx, cy = row['calq'][1:4]
        K4 = (f, f, cx, cy)
        ground = np.array([i not in solve.TOP_GATES for i in ids])
        res = solve.pnp_ransac(K4, solve.P64[ids], uv, ground)
        assert res is not None, row['seed']
        R, t = solve.lm_solver_pose(res[0], res[1], K4, solve.P64[ids], uv, 20000, 1e-5)
        want = _rmse(row['refine'][:3], row['refine'][3:], K4, solve.P64[ids], uv)
        got = _rmse(solve.log_so3(R), t, K4, solve.P64[ids], uv)
        assert abs(got - want) <= 1e-4 * want, (row['seed'], got, want)
        n += 1
    assert n > 0


# the same four checks for both runs of the driver (module docstring)
_CHECKS = (_refine_pose_follows_solvePnPRefineLM, _calibrate_planes_follows_calibrateCamera,
           _homography_ransac_agrees_with_findHomography_on_the_inliers, _pnp_and_refine_reach_the_same_reprojection_error)


@pytest.mark.parametrize('check', _CHECKS, ids=lambda f: f.__name__.lstrip('_'))
def test_oracle_against_opencv(check):
    check()


@pytest.mark.gpu
@pytest.mark.parametrize('check', _CHECKS, ids=lambda f: f.__name__.lstrip('_'))
def test_oracle_against_opencv_on_the_gpu_box(check):
    check()
