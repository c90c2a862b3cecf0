"""Oracle: pinhole camera model arithmetic (numpy fp64).  TEST INFRASTRUCTURE ONLY.

Follows /root/reference/baseline/camera.py:
  pan_tilt_roll_to_orientation :7-28      rotation_matrix_to_pan_tilt_roll :31-58
  to_json_parameters :156-175             from_json_parameters :177-218
  distort :220-247 (fp32 cast at :247)    project_point :249-268
  projection_rmse :270-277 (MEAN L2)      estimate_calibration_matrix_from_plane_homography :366-426
  from_homography :121-154
and /root/reference/src/models/hrnet/prediction.py:469-484 (good_camera / is_good_camera).
Pinned by tests/golden/camera.npz (captured from the imported reference Camera; numpy-only paths).
"""
import numpy as np


def pan_tilt_roll_to_orientation(pan, tilt, roll):
    Rpan = np.array([[np.cos(pan), -np.sin(pan), 0], [np.sin(pan), np.cos(pan), 0], [0, 0, 1]])
    Rroll = np.array([[np.cos(roll), -np.sin(roll), 0], [np.sin(roll), np.cos(roll), 0], [0, 0, 1]])
    Rtilt = np.array([[1, 0, 0], [0, np.cos(tilt), -np.sin(tilt)], [0, np.sin(tilt), np.cos(tilt)]])
    return Rpan @ (Rtilt @ Rroll)


def rotation_from_ptr(pan, tilt, roll):
    """camera.py:207-208 (the hand-expanded matrix at :198-205 is overwritten)."""
    return pan_tilt_roll_to_orientation(pan, tilt, roll).T


def rotation_to_ptr(rotation):
    """camera.py:31-58: ZXZ decomposition, solution with the smaller |roll|."""
    o = rotation.T
    t1 = np.arccos(o[2, 2])
    t2 = -t1
    s1 = 1.0 if np.sin(t1) > 0.0 else -1.0
    s2 = 1.0 if np.sin(t2) > 0.0 else -1.0
    p1 = np.arctan2(s1 * o[0, 2], s1 * -o[1, 2])
    p2 = np.arctan2(s2 * o[0, 2], s2 * -o[1, 2])
    r1 = np.arctan2(s1 * o[2, 0], s1 * o[2, 1])
    r2 = np.arctan2(s2 * o[2, 0], s2 * o[2, 1])
    if np.fabs(r1) < np.fabs(r2):
        return p1, t1, r1
    return p2, t2, r2


def project_point(position, rotation, fx, fy, pp, point3d):
    """camera.py:249-268 with all-zero distortion: identity map but through an fp32 round trip."""
    rp = rotation @ (np.asarray(point3d, dtype=np.float64) - position)
    if rp[2] <= 1e-3:
        return np.zeros(3)
    rp = rp / rp[2]
    d = np.array([rp[0], rp[1]], dtype=np.float32)          # distort() returns float32 (:247)
    # float32 * np.float64 focal length (cv2 matrices are float64) -> float64 arithmetic from here on
    return np.array([float(d[0]) * float(fx) + pp[0], float(d[1]) * float(fy) + pp[1], 1.0])


def projection_rmse(position, rotation, fx, fy, pp, pts3d, pts2d):
    """camera.py:270-277: MEAN of the per-point L2 distances (not a root-mean-square)."""
    proj = np.stack([project_point(position, rotation, fx, fy, pp, p)[:2] for p in pts3d])
    return float(np.mean(np.linalg.norm(np.asarray(pts2d, dtype=np.float64) - proj, axis=-1)))


def to_json(position, rotation, fx, fy, pp):
    pan, tilt, roll = rotation_to_ptr(rotation)
    return {
        'pan_degrees': pan * 180. / np.pi, 'tilt_degrees': tilt * 180. / np.pi,
        'roll_degrees': roll * 180. / np.pi, 'position_meters': list(map(float, position)),
        'x_focal_length': fx, 'y_focal_length': fy, 'principal_point': [pp[0], pp[1]],
        'radial_distortion': [0.0] * 6, 'tangential_distortion': [0.0] * 2,
        'thin_prism_distortion': [0.0] * 4,
    }


def k_from_plane_homography(H, pp=(480.0, 270.0)):
    """camera.py:366-426.  Returns (success, fx, fy) -- the principal point is forced to `pp`."""
    h = np.reshape(H, (9,))
    A = np.zeros((5, 6))
    A[0, 1] = 1.
    A[1, 0] = 1.
    A[1, 2] = -1.
    A[2, 3] = pp[1] / pp[0]
    A[2, 4] = -1.0
    A[3] = [h[0] * h[1], h[0] * h[4] + h[1] * h[3], h[3] * h[4],
            h[0] * h[7] + h[1] * h[6], h[3] * h[7] + h[4] * h[6], h[6] * h[7]]
    A[4] = [h[0] * h[0] - h[1] * h[1], 2 * h[0] * h[3] - 2 * h[1] * h[4], h[3] * h[3] - h[4] * h[4],
            2 * h[0] * h[6] - 2 * h[1] * h[7], 2 * h[3] * h[6] - 2 * h[4] * h[7], h[6] * h[6] - h[7] * h[7]]
    _, _, vh = np.linalg.svd(A)
    w = vh[-1]
    Wm = np.array([[w[0], w[1], w[3]], [w[1], w[2], w[4]], [w[3], w[4], w[5]]]) / w[5]
    try:
        Ktinv = np.linalg.cholesky(Wm)
    except np.linalg.LinAlgError:
        return False, 1.0, 1.0
    K = np.linalg.inv(Ktinv.T)
    K /= K[2, 2]
    return True, K[0, 0], K[1, 1]


def good_camera(fx, pos):
    """prediction.py:469-484."""
    return bool(10 <= fx <= 20000 and -250 < pos[0] < 250 and -250 < pos[1] < 250 and -100 < pos[2] < 0)
