"""Camera model behind the reference's ``baseline.camera.Camera`` surface (/root/reference/baseline/camera.py:77-426).

Public attributes (position, rotation, calibration, radial_distortion, thin_prism_disto, tangential_disto,
image_width, image_height, xfocal_length, yfocal_length, principal_point), method names, argument meaning and
the ten JSON keys are the reference's -- that is the API ``make_submit.py`` / ``evaluate_camera.py`` bind to.
The arithmetic underneath is this build's own:

* projection is vectorised over an (N,3) block of world points (``project_points``); ``project_point`` is its
  one-row form.  It keeps the reference's observable numerics: the z <= 1e-3 cut, and the float32 round trip of
  the normalised image coordinates that ``distort`` performs (camera.py:247) before the focal length is applied;
* the intrinsics from one plane homography use the image-of-the-absolute-conic constraints in closed form: with
  zero skew, square pixels and the principal-point direction fixed, omega has three unknowns and the two
  homography constraints give its null direction as a cross product -- no SVD, and the Cholesky factor of a
  matrix with this sparsity is three square roots;
* pan / tilt / roll are the ZXZ Euler angles of rotation^T, both tilt branches evaluated, the branch with the
  smaller |roll| kept (the reference's selection rule, camera.py:56-58).

``solve_pnp`` / ``refine_camera`` run on the GPU through libsncal.so (csrc/solve.hip).  Pinned by
tests/golden/camera.npz (captured from the imported reference).  ``draw_*`` is out of scope.
"""
import numpy as np

from . import _lib


def _rot_z(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]])


def _rot_x(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[1.0, 0.0, 0.0], [0.0, c, -s], [0.0, s, c]])


def pan_tilt_roll_to_orientation(pan, tilt, roll):
    """Orientation = Rz(pan) . Rx(tilt) . Rz(roll)   (camera.py:7-28; the camera's rotation is its transpose)."""
    return _rot_z(pan) @ (_rot_x(tilt) @ _rot_z(roll))


def rotation_matrix_to_pan_tilt_roll(rotation):
    """Inverse of the above.  cos(tilt) = O[2,2] leaves tilt = +-acos; each sign fixes pan and roll through the
    last column / last row of O.  Returns the branch with the smaller |roll| (ties: the negative-tilt one)."""
    O = np.asarray(rotation).T
    t_pos = np.arccos(O[2, 2])
    branches = []
    for tilt in (t_pos, -t_pos):
        sg = 1.0 if np.sin(tilt) > 0.0 else -1.0
        pan = np.arctan2(sg * O[0, 2], sg * -O[1, 2])
        roll = np.arctan2(sg * O[2, 0], sg * O[2, 1])
        branches.append((pan, tilt, roll))
    first, second = branches
    return first if np.fabs(first[2]) < np.fabs(second[2]) else second


def unproject_image_point(homography, point2D):
    q = np.linalg.solve(np.asarray(homography, dtype=np.float64), np.asarray(point2D, dtype=np.float64))
    return q / q[2]


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
        self.image_width, self.image_height = iwidth, iheight
        self.position, self.rotation, self.calibration = np.zeros(3), np.eye(3), np.eye(3)
        for name, size in (('radial_distortion', 6), ('thin_prism_disto', 4), ('tangential_disto', 2)):
            setattr(self, name, np.zeros(size))
        self.xfocal_length = self.yfocal_length = 1
        self.principal_point = (iwidth / 2, iheight / 2)

    def _set_intrinsics(self, fx, fy, pp):
        self.xfocal_length, self.yfocal_length, self.principal_point = fx, fy, pp
        K = np.eye(3)
        K[0, 0], K[1, 1], K[0, 2], K[1, 2] = fx, fy, pp[0], pp[1]
        self.calibration = K

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

    # ---- intrinsics / pose from a ground-plane homography (camera.py:121-154, 366-426) ---------------------
    def estimate_calibration_matrix_from_plane_homography(self, homography):
        """K from H = K [r1 r2 t].  omega = K^-T K^-1 restricted to zero skew (w01 = 0), square pixels (w00 = w11)
        and a principal point on the line through (cx, cy) (w12 = (cy/cx) w02) is  a*diag(1,1,0) + d*S + e*E33  with
        S the symmetric matrix carrying (1, cy/cx) in its last row/column.  r1 . r2 = 0 and |r1| = |r2| are two linear
        equations in (a, d, e); their solution is the cross product of the coefficient rows."""
        Hm = np.asarray(homography, dtype=np.float64).reshape(3, 3)
        h1, h2 = Hm[:, 0], Hm[:, 1]
        ratio = self.principal_point[1] / self.principal_point[0]

        def coeffs(u, v):     # u^T omega v as a row over (a, d, e)
            return np.array([u[0] * v[0] + u[1] * v[1],
                             u[0] * v[2] + u[2] * v[0] + ratio * (u[1] * v[2] + u[2] * v[1]),
                             u[2] * v[2]])
        a, d, e = np.cross(coeffs(h1, h2), coeffs(h1, h1) - coeffs(h2, h2))
        if e == 0.0 or not np.isfinite(a / e) or not np.isfinite(d / e):
            return False, np.eye(3)
        A, D = a / e, d / e
        # Cholesky of [[A,0,D],[0,A,rD],[D,rD,1]]: l00 = l11 = sqrt(A), l20 = D/l00, l21 = rD/l11, l22^2 = 1 - l20^2 - l21^2
        rem = 1.0 - (D * D) * (1.0 + ratio * ratio) / A if A > 0.0 else -1.0
        if not (A > 0.0 and rem > 0.0):
            return False, np.eye(3)          # omega is not positive definite: no real camera behind this homography
        focal = np.sqrt(rem / A)             # (K^-T)^-1 normalised to K[2,2] = 1: K00 = K11 = l22 / l00
        K = np.array([[focal, 0.0, -D / A], [0.0, focal, -ratio * D / A], [0.0, 0.0, 1.0]])
        self._set_intrinsics(K[0, 0], K[1, 1], (self.image_width / 2, self.image_height / 2))
        return True, K

    def from_homography(self, homography):
        ok, _ = self.estimate_calibration_matrix_from_plane_homography(homography)
        if not ok:
            return False
        M = np.linalg.inv(self.calibration) @ np.asarray(homography, dtype=np.float64)
        s1, s2 = 1.0 / np.linalg.norm(M[:, 0]), 1.0 / np.linalg.norm(M[:, 1])
        c1, c2 = M[:, 0] * s1, M[:, 1] * s2
        # nearest rotation to [c1 c2 c1xc2] in the Frobenius sense
        U, _, Vt = np.linalg.svd(np.stack([c1, c2, np.cross(c1, c2)], axis=1))
        if np.linalg.det(U @ Vt) < 0:
            U[:, 2] = -U[:, 2]
        self.rotation = U @ Vt
        self.position = -(self.rotation.T @ (M[:, 2] * np.sqrt(s1 * s2)))
        return True

    # ---- JSON (camera.py:156-218; the 10 keys and their order are the file format) ----------------------------
    def to_json_parameters(self):
        angles = [a * 180. / np.pi for a in rotation_matrix_to_pan_tilt_roll(self.rotation)]
        fields = list(zip(("pan_degrees", "tilt_degrees", "roll_degrees"), angles))
        fields += [("position_meters", self.position.tolist()),
                   ("x_focal_length", self.xfocal_length), ("y_focal_length", self.yfocal_length),
                   ("principal_point", [self.principal_point[0], self.principal_point[1]])]
        fields += [(key, getattr(self, attr).tolist()) for key, attr in (("radial_distortion", "radial_distortion"),
                   ("tangential_distortion", "tangential_disto"), ("thin_prism_distortion", "thin_prism_disto"))]
        return dict(fields)          # insertion order = the reference's key order (camera.py:162-173)

    def from_json_parameters(self, calib_json_object):
        js = calib_json_object
        pp = js["principal_point"]
        self.image_width, self.image_height = 2 * pp[0], 2 * pp[1]
        self._set_intrinsics(js["x_focal_length"], js["y_focal_length"], pp)
        angles = [js[k] * np.pi / 180. for k in ("pan_degrees", "tilt_degrees", "roll_degrees")]
        self.rotation = pan_tilt_roll_to_orientation(*angles).T
        for attr, key in (("position", "position_meters"), ("radial_distortion", "radial_distortion"),
                          ("tangential_disto", "tangential_distortion"), ("thin_prism_disto", "thin_prism_distortion")):
            setattr(self, attr, np.array(js[key], dtype='float'))

    # ---- projection (camera.py:220-277) ------------------------------------------------------------------
    def _distort_block(self, xy):
        """(N,2) float64 normalised coordinates -> (N,2) float32: rational radial + tangential + thin-prism model."""
        x, y = xy[:, 0], xy[:, 1]
        r = np.sqrt(x * x + y * y)
        pw = np.stack([r ** 2, r ** 4, r ** 6], axis=0)                     # (3,N)
        k = self.radial_distortion
        gain = (1 + k[:3] @ pw) / (1 + k[3:6] @ pw)
        p1, p2 = self.tangential_disto
        s = self.thin_prism_disto
        xd = x * gain + 2 * p1 * x * y + p2 * (pw[0] + 2 * x ** 2) + s[0] * pw[0] + s[1] * pw[1]
        yd = y * gain + 2 * p2 * x * y + p1 * (pw[0] + 2 * y ** 2) + s[2] * pw[0] + s[3] * pw[1]
        return np.stack([xd, yd], axis=1).astype(np.float32)               # the reference returns float32 (:247)

    def distort(self, point):
        return self._distort_block(np.asarray(point, dtype=np.float64)[None, :2])[0]

    def project_points(self, points3D, distort=True):
        """(N,3) world points -> (N,3) rows [x_px, y_px, 1]; a point at depth <= 1e-3 gives a row of zeros."""
        P = np.atleast_2d(np.asarray(points3D, dtype=np.float64))
        cam = (P - self.position) @ np.asarray(self.rotation).T
        front = cam[:, 2] > 1e-3
        out = np.zeros((P.shape[0], 3))
        if front.any():
            nrm = cam[front, :2] / cam[front, 2:3]
            # float32 values times a float64 focal length: float64 arithmetic from here on (numpy 1.24 promotion)
            d = self._distort_block(nrm).astype(np.float64) if distort else nrm
            out[front, 0] = d[:, 0] * float(self.xfocal_length) + self.principal_point[0]
            out[front, 1] = d[:, 1] * float(self.yfocal_length) + self.principal_point[1]
            out[front, 2] = 1.0
        return out

    def project_point(self, point3D, distort=True):
        return self.project_points(np.asarray(point3D, dtype=np.float64).reshape(1, 3), distort)[0]

    def projection_rmse(self, matched_points):
        """MEAN of the per-point pixel distances (the reference's definition, not a root-mean-square)."""
        world = np.array([m[0] for m in matched_points], dtype=np.float64)
        seen = np.array([m[1] for m in matched_points], dtype=np.float64)
        return np.mean(np.linalg.norm(seen - self.project_points(world)[:, :2], axis=1))

    def scale_resolution(self, factor):
        self.image_width, self.image_height = self.image_width * factor, self.image_height * factor
        self._set_intrinsics(self.xfocal_length * factor, self.yfocal_length * factor,
                             (self.image_width / 2, self.image_height / 2))
