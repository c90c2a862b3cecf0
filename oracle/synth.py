"""Build-owned deterministic synthetic inputs.  TEST INFRASTRUCTURE ONLY.

Random weights give flat heatmaps and no shipped checkpoints exist, so the decode and the solve are
driven by synthetic cameras -> projected template keypoints -> Gaussian heatmaps, the recipe the
reference itself holds commented out at /root/reference/src/models/hrnet/metamodel.py:69-75 and
/root/reference/src/models/hrnet/loss.py:21-52, 81-87.
"""
import numpy as np

from . import camera_math as cm
from .pitch import pitch_points


def sample_camera(rng: np.random.Generator):
    """A plausible broadcast camera inside prediction.py:469-475's bounds.  Returns dict."""
    pan = np.deg2rad(rng.uniform(-45, 45))
    tilt = np.deg2rad(rng.uniform(62, 86))
    roll = np.deg2rad(rng.normal(0, 1.0))
    pos = np.array([rng.uniform(-35, 35), rng.uniform(50, 95), rng.uniform(-35, -10)])
    f = float(np.exp(rng.uniform(np.log(900), np.log(5000))))
    # aim roughly at the pitch: recompute pan so the optical axis hits a point near the pitch
    target = np.array([rng.uniform(-40, 40), rng.uniform(-20, 20), 0.0])
    d = target - pos
    pan = np.arctan2(d[0], -d[1])          # optical axis = (sin pan sin tilt, -cos pan sin tilt, cos tilt)
    tilt = np.arctan2(np.hypot(d[0], d[1]), d[2])
    R = cm.rotation_from_ptr(pan, tilt, roll)
    return {'position': pos, 'rotation': R, 'f': f, 'pp': (480.0, 270.0)}


def project_template(cam, img_wh=(960, 540)):
    """(57,2) projections + (57,) visibility (in front of the camera and inside the image)."""
    P = pitch_points()
    uv = np.zeros((57, 2))
    vis = np.zeros(57, dtype=bool)
    for i in range(57):
        q = cm.project_point(cam['position'], cam['rotation'], cam['f'], cam['f'], cam['pp'], P[i])
        if q[2] == 0:
            continue
        uv[i] = q[:2]
        vis[i] = (0 <= q[0] < img_wh[0]) and (0 <= q[1] < img_wh[1])
    return uv, vis


def synth_keypoints(seed: int, sigma_px: float = 1.0, grid: float = 2.0, outlier_frac: float = 0.03,
                    min_visible: int = 0, img_wh=(960, 540)):
    """One frame's (57,3) float32 [x, y, conf] rows the way D1 would emit them, plus the camera.

    Visible points: projected + N(0, sigma) noise, snapped to the decode grid, conf~U(.55,1).
    Invisible points: (0,0) location with conf~U(0,.15).  A few outliers (random in-image location,
    conf > .5) exercise the RANSAC branches.  `min_visible` re-draws cameras until enough points show.
    """
    rng = np.random.Generator(np.random.PCG64(seed))
    while True:
        cam = sample_camera(rng)
        uv, vis = project_template(cam, img_wh)
        if vis.sum() >= min_visible:
            break
    kp = np.zeros((57, 3), dtype=np.float32)
    for i in range(57):
        if vis[i]:
            p = uv[i] + rng.normal(0, sigma_px, 2)
            if rng.random() < outlier_frac:
                p = np.array([rng.uniform(0, img_wh[0] - 1), rng.uniform(0, img_wh[1] - 1)])
            p = np.round(p / grid) * grid
            p[0] = min(max(p[0], 0), img_wh[0] - grid)
            p[1] = min(max(p[1], 0), img_wh[1] - grid)
            kp[i] = (p[0], p[1], rng.uniform(0.55, 1.0))
        else:
            kp[i] = (0.0, 0.0, rng.uniform(0.0, 0.15))
    return kp, cam


def create_target(kp: np.ndarray, sigma: float, hw) -> np.ndarray:
    """HRNetLoss.create_target, loss.py:81-87 with the reference's visibility test (loss.py:49: any component == 1);
    kp (B,N,3) [x, y, vis] in heatmap pixels.  exp is float32(exp(float64)), the correctly rounded value."""
    kp = np.asarray(kp, dtype=np.float32)
    h, w = hw
    sig = np.float32(sigma)

    def g(r, mu):
        d = ((r - mu[..., None]) / sig).astype(np.float32)
        return np.exp((-(d * d).astype(np.float32) / np.float32(2.0)).astype(np.float64)).astype(np.float32)
    gx = g(np.arange(w, dtype=np.float32), kp[..., 0])
    gy = g(np.arange(h, dtype=np.float32), kp[..., 1])
    hm = (gx[:, :, None, :] * gy[:, :, :, None]).astype(np.float32)
    vis = np.any(kp == np.float32(1.0), axis=-1)
    hm = np.where(vis[..., None, None], hm, np.float32(0)).astype(np.float32)
    bg = (np.float32(1.0) - hm.max(axis=1, keepdims=True)).astype(np.float32)
    return np.concatenate([hm, bg], axis=1)


def gaussian_heatmaps(kp_hm: np.ndarray, visible: np.ndarray, sigma: float, hw) -> np.ndarray:
    """loss.py:7-52 + :81-87: (B,N,2) keypoints in heatmap units -> (B,N+1,h,w) fp32 heatmaps
    (amplitude-1 separable Gaussians; last channel = 1 - max over keypoint channels)."""
    h, w = hw
    xr = np.arange(w, dtype=np.float32)
    yr = np.arange(h, dtype=np.float32)
    x = kp_hm[..., 0].astype(np.float32)[..., None]
    y = kp_hm[..., 1].astype(np.float32)[..., None]
    gx = np.exp(-(((xr - x) / np.float32(sigma)) ** 2) / np.float32(2.0)).astype(np.float32)
    gy = np.exp(-(((yr - y) / np.float32(sigma)) ** 2) / np.float32(2.0)).astype(np.float32)
    hm = np.einsum('bnw,bnh->bnhw', gx, gy).astype(np.float32)
    hm = np.where(visible[..., None, None], hm, np.float32(0)).astype(np.float32)
    bg = (np.float32(1.0) - hm.max(axis=1, keepdims=True)).astype(np.float32)
    return np.concatenate([hm, bg], axis=1)


def synth_logp(seeds, hw=(270, 480), img_wh=(960, 540), sigma=2.0, floor=1e-12):
    """(B,58,h,w) fp32 log-heatmaps for a list of seeds + the (B,57,3) keypoints that made them."""
    h, w = hw
    sx, sy = img_wh[0] / w, img_wh[1] / h
    kps = np.stack([synth_keypoints(s, grid=sx)[0] for s in seeds])
    vis = kps[..., 2] > 0.5
    kp_hm = np.stack([kps[..., 0] / sx, kps[..., 1] / sy], axis=-1)
    hm = gaussian_heatmaps(kp_hm, vis, sigma, hw)
    # scale each visible channel by its confidence so conf survives the decode
    hm[:, :57] *= kps[..., 2][..., None, None].astype(np.float32)
    with np.errstate(divide='ignore'):
        logp = np.log(np.maximum(hm, np.float32(floor)) if floor else hm).astype(np.float32)
    return logp, kps
