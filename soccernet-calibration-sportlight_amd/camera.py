"""Camera model -- host mirror of /root/reference/baseline/camera.py:77-426 (class Camera).

Same public attributes (position, rotation, calibration, radial_distortion, thin_prism_disto, tangential_disto,
image_width, image_height, xfocal_length, yfocal_length, principal_point) and methods (solve_pnp,
refine_camera, from_homography, to_json_parameters, from_json_parameters, distort, project_point,
projection_rmse, scale_resolution, estimate_calibration_matrix_from_plane_homography).  Plain numpy arrays,
mutated in place, picklable.  The two optimisation entry points (solve_pnp, refine_camera) run on the GPU
through libsncal.so (csrc/solve.hip); everything else is small fp64 host arithmetic.  draw_* (visualisation)
is out of scope.
"""
import numpy as np

from . import _lib


def pan_tilt_roll_to_orientation(pan, tilt, roll):
    Rpan = np.array([[np.cos(pan), -np.sin(pan), 0], [np.sin(pan), np.cos(pan), 0], [0, 0, 1]])
    Rroll = np.array([[np.cos(roll), -np.sin(roll), 0], [np.sin(roll), np.cos(roll), 0], [0, 0, 1]])
    Rtilt = np.array([[1, 0, 0], [0, np.cos(tilt), -np.sin(tilt)], [0, np.sin(tilt), np.cos(tilt)]])
    return np.dot(Rpan, np.dot(Rtilt, Rroll))


def rotation_matrix_to_pan_tilt_roll(rotation):
    """ZXZ Euler decomposition; of the two solutions the one with the smaller |roll| (camera.py:31-58)."""
    orientation = np.transpose(rotation)
    first_tilt = np.arccos(orientation[2, 2])
    second_tilt = -first_tilt
    s1 = 1. if np.sin(first_tilt) > 0. else -1.
    s2 = 1. if np.sin(second_tilt) > 0. else -1.
    first_pan = np.arctan2(s1 * orientation[0, 2], s1 * -orientation[1, 2])
    second_pan = np.arctan2(s2 * orientation[0, 2], s2 * -orientation[1, 2])
    first_roll = np.arctan2(s1 * orientation[2, 0], s1 * orientation[2, 1])
    second_roll = np.arctan2(s2 * orientation[2, 0], s2 * orientation[2, 1])
    if np.fabs(first_roll) < np.fabs(second_roll):
        return first_pan, first_tilt, first_roll
    return second_pan, second_tilt, second_roll


def unproject_image_point(homography, point2D):
    pitchpoint = np.linalg.inv(homography) @ point2D
    return pitchpoint / pitchpoint[2]


def _gpu_pnp(mode, calibration, rotation, position, point_matches, max_iters=0, eps=0.0):
    """One-frame call of sncal_solve_pnp (mode 1) / sncal_pnp_refine_lm (mode 0)."""
    import torch
    L = _lib.lib()
    if not torch.cuda.is_available():
        raise _lib.SncalError('Camera.solve_pnp / refine_camera need a GPU (libsncal.so has no CPU path)')
    obj = np.array([pt[0] for pt in point_matches], dtype=np.float64).reshape(-1, 3)
    img = np.array([pt[1] for pt in point_matches], dtype=np.float64).reshape(-1, 2)
    n = obj.shape[0]
    if n > 64:
        raise _lib.SncalError('at most 64 point matches per camera')
    dev = torch.device('cuda', torch.cuda.current_device())
    K = np.array([calibration[0, 0], calibration[1, 1], calibration[0, 2], calibration[1, 2]], dtype=np.float64)
    d_K = torch.from_numpy(K).to(dev)
    d_o = torch.from_numpy(np.ascontiguousarray(obj)).to(dev)
    d_i = torch.from_numpy(np.ascontiguousarray(img)).to(dev)
    d_n = torch.tensor([n], dtype=torch.int32, device=dev)
    rt = np.concatenate([np.asarray(rotation, dtype=np.float64).reshape(9), np.asarray(position, dtype=np.float64).reshape(3)])
    d_rt = torch.from_numpy(rt).to(dev)
    d_rm = torch.full((1,), -2.0, dtype=torch.float64, device=dev)
    s = _lib.current_stream_ptr()
    if mode == 1:
        _lib.check(L.sncal_solve_pnp(d_K.data_ptr(), d_o.data_ptr(), d_i.data_ptr(), d_n.data_ptr(), 1, n,
                                     d_rt.data_ptr(), s), 'sncal_solve_pnp')
        d_rm = None
    else:
        _lib.check(L.sncal_pnp_refine_lm(d_K.data_ptr(), d_o.data_ptr(), d_i.data_ptr(), d_n.data_ptr(), 1, n,
                                         d_rt.data_ptr(), d_rm.data_ptr(), int(max_iters), float(eps), s),
                   'sncal_pnp_refine_lm')
    out = d_rt.cpu().numpy()
    changed = not np.array_equal(out, rt)
    return out[:9].reshape(3, 3).copy(), out[9:].copy(), changed


class Camera:
    def __init__(self, iwidth=960, iheight=540):
        self.position = np.zeros(3)
        self.rotation = np.eye(3)
        self.calibration = np.eye(3)
        self.radial_distortion = np.zeros(6)
        self.thin_prism_disto = np.zeros(4)
        self.tangential_disto = np.zeros(2)
        self.image_width = iwidth
        self.image_height = iheight
        self.xfocal_length = 1
        self.yfocal_length = 1
        self.principal_point = (self.image_width / 2, self.image_height / 2)

    # ---- GPU-backed optimisation (S6, S7) -----------------------------------------------------------
    def solve_pnp(self, point_matches):
        """cv.solvePnPRansac + Rodrigues of the reference (camera.py:92-103)."""
        R, pos, changed = _gpu_pnp(1, self.calibration, self.rotation, self.position, point_matches)
        if not changed:
            raise _lib.SncalError('solve_pnp: no pose with >= 4 inliers')
        self.rotation, self.position = R, pos

    def refine_camera(self, pointMatches):
        """cv.solvePnPRefineLM of the reference (camera.py:105-119): 6-DoF LM, K fixed, to convergence."""
        R, pos, _ = _gpu_pnp(0, self.calibration, self.rotation, self.position, pointMatches)
        self.rotation, self.position = R, pos

    # ---- host arithmetic --------------------------------------------------------------------------------
    def from_homography(self, homography):
        success, _ = self.estimate_calibration_matrix_from_plane_homography(homography)
        if not success:
            return False
        hprim = np.linalg.inv(self.calibration) @ homography
        lambda1 = 1 / np.linalg.norm(hprim[:, 0])
        lambda2 = 1 / np.linalg.norm(hprim[:, 1])
        lambda3 = np.sqrt(lambda1 * lambda2)
        r0 = hprim[:, 0] * lambda1
        r1 = hprim[:, 1] * lambda2
        R = np.column_stack((r0, r1, np.cross(r0, r1)))
        u, s, vh = np.linalg.svd(R)
        R = u @ vh
        if np.linalg.det(R) < 0:
            u[:, 2] *= -1
            R = u @ vh
        self.rotation = R
        self.position = -np.transpose(R) @ (hprim[:, 2] * lambda3)
        return True

    def to_json_parameters(self):
        pan, tilt, roll = rotation_matrix_to_pan_tilt_roll(self.rotation)
        return {
            "pan_degrees": pan * 180. / np.pi,
            "tilt_degrees": tilt * 180. / np.pi,
            "roll_degrees": roll * 180. / np.pi,
            "position_meters": self.position.tolist(),
            "x_focal_length": self.xfocal_length,
            "y_focal_length": self.yfocal_length,
            "principal_point": [self.principal_point[0], self.principal_point[1]],
            "radial_distortion": self.radial_distortion.tolist(),
            "tangential_distortion": self.tangential_disto.tolist(),
            "thin_prism_distortion": self.thin_prism_disto.tolist(),
        }

    def from_json_parameters(self, calib_json_object):
        self.principal_point = calib_json_object["principal_point"]
        self.image_width = 2 * self.principal_point[0]
        self.image_height = 2 * self.principal_point[1]
        self.xfocal_length = calib_json_object["x_focal_length"]
        self.yfocal_length = calib_json_object["y_focal_length"]
        self.calibration = np.array([[self.xfocal_length, 0, self.principal_point[0]],
                                     [0, self.yfocal_length, self.principal_point[1]], [0, 0, 1]], dtype='float')
        pan = calib_json_object['pan_degrees'] * np.pi / 180.
        tilt = calib_json_object['tilt_degrees'] * np.pi / 180.
        roll = calib_json_object['roll_degrees'] * np.pi / 180.
        self.rotation = np.transpose(pan_tilt_roll_to_orientation(pan, tilt, roll))
        self.position = np.array(calib_json_object['position_meters'], dtype='float')
        self.radial_distortion = np.array(calib_json_object['radial_distortion'], dtype='float')
        self.tangential_disto = np.array(calib_json_object['tangential_distortion'], dtype='float')
        self.thin_prism_disto = np.array(calib_json_object['thin_prism_distortion'], dtype='float')

    def distort(self, point):
        numerator = 1
        denominator = 1
        radius = np.sqrt(point[0] * point[0] + point[1] * point[1])
        for i in range(3):
            numerator += self.radial_distortion[i] * radius ** (2 * (i + 1))
            denominator += self.radial_distortion[i + 3] * radius ** (2 * (i + 1))
        f = numerator / denominator
        xpp = point[0] * f + 2 * self.tangential_disto[0] * point[0] * point[1] + \
            self.tangential_disto[1] * (radius ** 2 + 2 * point[0] ** 2) + \
            self.thin_prism_disto[0] * radius ** 2 + self.thin_prism_disto[1] * radius ** 4
        ypp = point[1] * f + 2 * self.tangential_disto[1] * point[0] * point[1] + \
            self.tangential_disto[0] * (radius ** 2 + 2 * point[1] ** 2) + \
            self.thin_prism_disto[2] * radius ** 2 + self.thin_prism_disto[3] * radius ** 4
        return np.array([xpp, ypp], dtype=np.float32)          # float32, like the reference (:247)

    def project_point(self, point3D, distort=True):
        point = point3D - self.position
        rotated_point = self.rotation @ np.transpose(point)
        if rotated_point[2] <= 1e-3:
            return np.zeros(3)
        rotated_point = rotated_point / rotated_point[2]
        d = self.distort(rotated_point) if distort else rotated_point
        # float32 * float64 focal length -> float64, the promotion of the reference's pinned numpy 1.24
        x = float(d[0]) * float(self.xfocal_length) + self.principal_point[0]
        y = float(d[1]) * float(self.yfocal_length) + self.principal_point[1]
        return np.array([x, y, 1])

    def projection_rmse(self, matched_points):
        target_pts = np.array([pt[0] for pt in matched_points])
        img_pts = np.array([pt[1] for pt in matched_points])
        projected = np.stack([self.project_point(p3d)[:2] for p3d in target_pts], axis=0)
        return np.mean(np.linalg.norm(img_pts - projected, ord=2.0, axis=-1))

    def scale_resolution(self, factor):
        self.xfocal_length = self.xfocal_length * factor
        self.yfocal_length = self.yfocal_length * factor
        self.image_width = self.image_width * factor
        self.image_height = self.image_height * factor
        self.principal_point = (self.image_width / 2, self.image_height / 2)
        self.calibration = np.array([[self.xfocal_length, 0, self.principal_point[0]],
                                     [0, self.yfocal_length, self.principal_point[1]], [0, 0, 1]], dtype='float')

    def estimate_calibration_matrix_from_plane_homography(self, homography):
        """Image of the absolute conic from one plane homography (camera.py:366-426, HZ alg. 8.2)."""
        H = np.reshape(homography, (9,))
        A = np.zeros((5, 6))
        A[0, 1] = 1.
        A[1, 0] = 1.
        A[1, 2] = -1.
        A[2, 3] = self.principal_point[1] / self.principal_point[0]
        A[2, 4] = -1.0
        A[3] = [H[0] * H[1], H[0] * H[4] + H[1] * H[3], H[3] * H[4], H[0] * H[7] + H[1] * H[6],
                H[3] * H[7] + H[4] * H[6], H[6] * H[7]]
        A[4] = [H[0] * H[0] - H[1] * H[1], 2 * H[0] * H[3] - 2 * H[1] * H[4], H[3] * H[3] - H[4] * H[4],
                2 * H[0] * H[6] - 2 * H[1] * H[7], 2 * H[3] * H[6] - 2 * H[4] * H[7], H[6] * H[6] - H[7] * H[7]]
        _, _, vh = np.linalg.svd(A)
        w = vh[-1]
        W = np.array([[w[0], w[1], w[3]], [w[1], w[2], w[4]], [w[3], w[4], w[5]]]) / w[5]
        try:
            Ktinv = np.linalg.cholesky(W)
        except np.linalg.LinAlgError:
            return False, np.eye(3)
        K = np.linalg.inv(np.transpose(Ktinv))
        K /= K[2, 2]
        self.xfocal_length = K[0, 0]
        self.yfocal_length = K[1, 1]
        self.principal_point = (self.image_width / 2, self.image_height / 2)
        self.calibration = np.array([[self.xfocal_length, 0, self.principal_point[0]],
                                     [0, self.yfocal_length, self.principal_point[1]], [0, 0, 1]], dtype='float')
        return True, K
