"""Synthetic workload generator for the solve stage (benchmarks / smoke runs).

No trained checkpoints ship with the reference and random weights give flat heatmaps, so end-to-end runs
drive the camera solve with keypoints obtained by projecting the pitch template through plausible broadcast
cameras -- the trick the reference itself holds (commented out) at
/root/reference/src/models/hrnet/metamodel.py:69-75.  Rows look like HRNetPredictionTransform output:
[x_px, y_px, conf] on the 2-px decode grid of a 270x480 heatmap.
"""
import numpy as np

from .camera import Camera, pan_tilt_roll_to_orientation
from .pitch import PITCH_ARRAY


def random_camera(rng: np.random.Generator) -> Camera:
    cam = Camera(960, 540)
    pos = np.array([rng.uniform(-35, 35), rng.uniform(50, 95), rng.uniform(-35, -10)])
    target = np.array([rng.uniform(-40, 40), rng.uniform(-20, 20), 0.0])
    d = target - pos
    pan = np.arctan2(d[0], -d[1])
    tilt = np.arctan2(np.hypot(d[0], d[1]), d[2])
    roll = np.deg2rad(rng.normal(0, 1.0))
    f = float(np.exp(rng.uniform(np.log(900), np.log(5000))))
    cam.position = pos
    cam.rotation = np.transpose(pan_tilt_roll_to_orientation(pan, tilt, roll))
    cam.xfocal_length = cam.yfocal_length = np.float64(f)
    cam.calibration = np.array([[f, 0, 480.0], [0, f, 270.0], [0, 0, 1.0]])
    return cam


def keypoints_for_camera(cam: Camera, rng: np.random.Generator, sigma_px=1.0, outlier_frac=0.03, grid=2.0):
    kp = np.zeros((57, 3), dtype=np.float32)
    for i in range(57):
        q = cam.project_point(PITCH_ARRAY[i])
        vis = q[2] != 0 and 0 <= q[0] < 960 and 0 <= q[1] < 540
        if vis:
            p = q[:2] + rng.normal(0, sigma_px, 2)
            if rng.random() < outlier_frac:
                p = np.array([rng.uniform(0, 959), rng.uniform(0, 539)])
            p = np.round(p / grid) * grid
            kp[i] = (min(max(p[0], 0), 960 - grid), min(max(p[1], 0), 540 - grid), rng.uniform(0.55, 1.0))
        else:
            kp[i] = (0.0, 0.0, rng.uniform(0.0, 0.15))
    return kp


def synthetic_keypoints(n: int, seed: int = 0, min_visible: int = 8) -> np.ndarray:
    """(n,57,3) float32 keypoint rows; cameras are re-drawn until at least `min_visible` points show."""
    rng = np.random.Generator(np.random.PCG64(seed))
    out = np.zeros((n, 57, 3), dtype=np.float32)
    for b in range(n):
        while True:
            kp = keypoints_for_camera(random_camera(rng), rng)
            if (kp[:, 2] > 0.5).sum() >= min_visible:
                break
        out[b] = kp
    return out
