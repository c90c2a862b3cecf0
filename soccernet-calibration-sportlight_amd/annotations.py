"""Annotation geometry (second half): SoccerNet line / circle annotations -> the 57 keypoint labels.

Host-side label preparation of the reference's training data path -- the counterpart of
    /root/reference/src/datatools/intersections.py:53-124   (intersection, get_intersections)
    /root/reference/src/datatools/ellipse.py:275-516        (tangent points, conic x line, side selection, homography fill)
    /root/reference/src/datatools/line.py:9-32, geom.py:33-60 (closest points, image bounds)
with the same entry point, ``get_intersections(points, img_size, within_image, margin) -> (labels, mask)``: `points` maps an
annotation class name to its polyline in normalised image coordinates, `labels` maps keypoint id 0..56 to (x, y) in pixels or
None, `mask` lists the circle-derived ids that could not be produced.  (The heatmap half of N4 is csrc/target.hip.)

The arithmetic is this build's own:
  * straight lines are ordinary least-squares fits y = k x + h from centred moments; the reference's "x = const" rule is kept
    as a decision (it changes which points enter the refinement), conics and lines meet in homogeneous coordinates;
  * the ellipse is a direct least-squares conic fit with the ellipse constraint 4ac - b^2 = 1 (Halir & Flusser's partitioned
    form) -- the algorithm behind the `lsq-ellipse` package the reference depends on;
  * tangent points from an external point P are the intersections of the conic with the POLAR LINE of P (C P); the two
    solutions are ordered like the reference's `idx` (by the slope of the tangent, through the sign of its closed-form C term);
  * conic x line is the conic restricted to the line's parametrisation (a quadratic in one parameter);
  * the missing circle points come from a RANSAC homography pitch -> image (normalised DLT, deterministic sampling).
No OpenCV, no lsq-ellipse.  Pinned by tests/golden/annotations.json: the imported reference functions (numpy-only parts) run on
synthetic annotations with the two third-party calls replaced by the fit / homography of this file (tools/make_golden.py).
"""
from typing import Dict, List, Optional, Tuple

import numpy as np

from .lines import LINE_INTERSECTIONS
from .pitch import INTERSECTON_TO_PITCH_POINTS, PITCH_POINTS

NOT_ON_PLANE = (0, 1, 24, 25)
_ID = {v: k for k, v in INTERSECTON_TO_PITCH_POINTS.items()}

# circle-derived keypoints: (annotation class) -> [(keypoint name, kind, argument, selector)]
#   tangent: argument = keypoint the tangent passes through, selector = which of the two tangents (reference idx)
#   cross:   argument = annotated line that cuts the circle,  selector = 'Top' / 'Bottom'
CONIC_POINTS = {
    'Circle central': [('CENTER_CIRCLE_TANGENT_TR', 'tangent', 'T_TOUCH_AND_HALFWAY_LINES_INTERSECTION', 0),
                       ('CENTER_CIRCLE_TANGENT_TL', 'tangent', 'T_TOUCH_AND_HALFWAY_LINES_INTERSECTION', 1),
                       ('CENTER_CIRCLE_TANGENT_BR', 'tangent', 'B_TOUCH_AND_HALFWAY_LINES_INTERSECTION', 0),
                       ('CENTER_CIRCLE_TANGENT_BL', 'tangent', 'B_TOUCH_AND_HALFWAY_LINES_INTERSECTION', 1),
                       ('T_HALFWAY_LINE_AND_CENTER_CIRCLE_INTERSECTION', 'cross', 'Middle line', 'Top'),
                       ('B_HALFWAY_LINE_AND_CENTER_CIRCLE_INTERSECTION', 'cross', 'Middle line', 'Bottom')],
    'Circle left': [('BL_16M_LINE_AND_PENALTY_ARC_INTERSECTION', 'cross', 'Big rect. left main', 'Bottom'),
                    ('TL_16M_LINE_AND_PENALTY_ARC_INTERSECTION', 'cross', 'Big rect. left main', 'Top'),
                    ('LEFT_CIRCLE_TANGENT_T', 'tangent', 'L_PENALTY_AREA_TR_CORNER', 0),
                    ('LEFT_CIRCLE_TANGENT_B', 'tangent', 'L_PENALTY_AREA_BR_CORNER', 1)],
    'Circle right': [('BR_16M_LINE_AND_PENALTY_ARC_INTERSECTION', 'cross', 'Big rect. right main', 'Bottom'),
                     ('TR_16M_LINE_AND_PENALTY_ARC_INTERSECTION', 'cross', 'Big rect. right main', 'Top'),
                     ('RIGHT_CIRCLE_TANGENT_T', 'tangent', 'R_PENALTY_AREA_TL_CORNER', 1),
                     ('RIGHT_CIRCLE_TANGENT_B', 'tangent', 'R_PENALTY_AREA_BL_CORNER', 0)],
}


# ---- lines -------------------------------------------------------------------------------------------------------
def is_vertical(pts: np.ndarray, ref: Optional[float] = None) -> bool:
    """The reference's "x = const" test (numpy.isclose with atol 0.5): every x within half a pixel of `ref` -- the mean x for
    the line x line case (intersections.py:70-73), the FIRST x for the circle x line case (ellipse.py:443-444)."""
    r = float(pts[:, 0].mean()) if ref is None else float(ref)
    return bool(np.all(np.abs(pts[:, 0] - r) <= 0.5 + 1e-5 * abs(r)))


def fit_slope(pts: np.ndarray) -> Tuple[float, float]:
    """Least-squares y = k x + h (ordinary regression of y on x, what numpy's polyfit of degree 1 minimises)."""
    x, y = pts[:, 0].astype(np.float64), pts[:, 1].astype(np.float64)
    xm, ym = x.mean(), y.mean()
    sxx = float(((x - xm) ** 2).sum())
    k = float(((x - xm) * (y - ym)).sum()) / sxx
    return k, ym - k * xm


def two_nearest(pts: np.ndarray, x: float, y: float, any_side: bool = False) -> Optional[np.ndarray]:
    """The annotated point nearest to (x, y) and a partner: simply the second nearest (`any_side`), or the nearest one such
    that (x, y) lies inside the pair's bounding box (None if there is none)."""
    order = np.argsort(np.hypot(pts[:, 0] - x, pts[:, 1] - y), kind='stable')
    first = pts[order[0]]
    if any_side:
        return np.vstack((first, pts[order[1]]))
    for j in order[1:]:
        lo, hi = np.minimum(first, pts[j]), np.maximum(first, pts[j])
        if lo[0] <= x <= hi[0] and lo[1] <= y <= hi[1]:
            return np.vstack((first, pts[j]))
    return None


def line_intersection(l1: np.ndarray, l2: np.ndarray) -> Optional[Tuple[float, float]]:
    """Intersection of two annotated polylines: fit, intersect, then repeat with the two points of each polyline nearest to
    the estimate (annotations bend with lens distortion; the local segment is the better line)."""
    eps = 1e-18
    while True:
        v1, v2 = is_vertical(l1), is_vertical(l2)
        if v1 and v2:
            return None
        if v1:
            x = float(l1[:, 0].mean())
            k, h = fit_slope(l2)
            y = k * x + h
        elif v2:
            x = float(l2[:, 0].mean())
            k, h = fit_slope(l1)
            y = k * x + h
        else:
            k1, h1 = fit_slope(l1)
            k2, h2 = fit_slope(l2)
            x = (h2 - h1) / (k1 - k2 + eps)
            y = k1 * x + h1
        if l1.shape[0] <= 2 and l2.shape[0] <= 2:
            return (x, y)
        l1, l2 = two_nearest(l1, x, y, True), two_nearest(l2, x, y, True)


def inside(point, img_size=(960, 540), within_img: bool = True, margin: float = 0.0):
    if point is None or not within_img:
        return point
    x, y = point
    return point if (-margin <= x <= img_size[0] + margin and -margin <= y <= img_size[1] + margin) else None


# ---- conics ------------------------------------------------------------------------------------------------------
def fit_ellipse(pts: np.ndarray) -> Optional[np.ndarray]:
    """Direct least-squares ellipse through >= 5 points: coefficients (a, b, c, d, e, f) of a x^2 + b x y + c y^2 + d x + e y + f = 0
    (any scale).  Partitioned normal equations with the constraint 4 a c - b^2 = 1 (Halir & Flusser 1998)."""
    x, y = pts[:, 0].astype(np.float64), pts[:, 1].astype(np.float64)
    if x.size < 5:
        return None
    D1 = np.stack([x * x, x * y, y * y], axis=1)
    D2 = np.stack([x, y, np.ones_like(x)], axis=1)
    S1, S2, S3 = D1.T @ D1, D1.T @ D2, D2.T @ D2
    try:
        T = -np.linalg.solve(S3, S2.T)
    except np.linalg.LinAlgError:
        return None
    M = S1 + S2 @ T
    M = np.stack([M[2] / 2.0, -M[1], M[0] / 2.0])           # premultiplication by the inverse constraint matrix
    w, v = np.linalg.eig(M)
    v = np.real(v)
    cond = 4.0 * v[0] * v[2] - v[1] ** 2
    ok = np.nonzero(cond > 0)[0]
    if ok.size == 0:
        return None
    a1 = v[:, ok[0]]
    return np.concatenate([a1, T @ a1])


def _conic_matrix(q) -> np.ndarray:
    a, b, c, d, e, f = q
    return np.array([[a, b / 2, d / 2], [b / 2, c, e / 2], [d / 2, e / 2, f]], dtype=np.float64)


def conic_line_points(q, p0: np.ndarray, direction: np.ndarray) -> List[np.ndarray]:
    """Intersections of the conic with the line p0 + t * direction (homogeneous 3-vectors): roots of a quadratic in t."""
    C = _conic_matrix(q)
    qa, qb, qc = direction @ C @ direction, 2.0 * (p0 @ C @ direction), p0 @ C @ p0
    if abs(qa) < 1e-300:
        return []
    disc = qb * qb - 4.0 * qa * qc
    if disc < 0:
        return []
    s = np.sqrt(disc)
    out = []
    for t in ((-qb - s) / (2 * qa), (-qb + s) / (2 * qa)):
        h = p0 + t * direction
        out.append(h[:2] / h[2])
    return out


def tangent_points(q, p: Tuple[float, float]) -> Optional[Tuple[np.ndarray, np.ndarray]]:
    """The two points where the tangents from the external point p touch the conic, in the reference's order: index 0 is the
    tangent of slope (A - B) / C, index 1 the one of slope (A + B) / C in the reference's closed form, i.e. the smaller slope
    first when C > 0 and the larger first when C < 0 (B >= 0)."""
    a, b, c, d, e, f = q
    x0, y0 = float(p[0]), float(p[1])
    polar = _conic_matrix(q) @ np.array([x0, y0, 1.0])               # every tangency point lies on C p
    l0, l1, l2 = polar
    if abs(l0) >= abs(l1):                                           # a point on the polar line and its direction
        p0, direction = np.array([-l2 / l0, 0.0, 1.0]), np.array([-l1, l0, 0.0])
    else:
        p0, direction = np.array([0.0, -l2 / l1, 1.0]), np.array([-l1, l0, 0.0])
    pts = conic_line_points(q, p0, direction)
    if len(pts) != 2:
        return None
    slope = [(t[1] - y0) / (t[0] - x0) for t in pts]
    csign = 4 * a * c * x0 ** 2 - b ** 2 * x0 ** 2 - 2 * b * e * x0 + 4 * c * d * x0 + 4 * c * f - e ** 2
    first_is_smaller = csign > 0
    if (slope[0] < slope[1]) != first_is_smaller:
        pts = pts[::-1]
    return pts[0], pts[1]


def conic_cross_line(q, line: np.ndarray) -> Optional[np.ndarray]:
    """The two intersections of the fitted ellipse with an annotated polyline, each refined with the polyline's local segment
    (the two annotated points that bracket the first estimate).  (2,2) array or None."""
    def cut(poly):
        if is_vertical(poly, poly[0, 0]):
            xv = float(poly[0, 0])                                   # (the reference "averages" the first x only)
            return conic_line_points(q, np.array([xv, 0.0, 1.0]), np.array([0.0, 1.0, 0.0]))
        k, h = fit_slope(poly)
        return conic_line_points(q, np.array([0.0, h, 1.0]), np.array([1.0, k, 0.0]))
    first = cut(line)
    if len(first) != 2:
        return None
    first = sorted(first, key=lambda t: (t[0], t[1]))       # any fixed order: the consumer (`pick_side`) is symmetric in the two
    out = []
    for pt in first:
        seg = two_nearest(line, pt[0], pt[1])
        if seg is not None and abs(seg[0, 0] - seg[1, 0]) > 0:
            k, h = fit_slope(seg)
            cand = conic_line_points(q, np.array([0.0, h, 1.0]), np.array([1.0, k, 0.0]))
            if cand:
                pt = min(cand, key=lambda c: (c[0] - pt[0]) ** 2 + (c[1] - pt[1]) ** 2)
        out.append(pt)
    return np.array(out)


def pick_side(pair: np.ndarray, points: Dict[str, List], img_size, circle: str, line: str, side: str) -> np.ndarray:
    """Which of the two circle x line intersections is the 'Top' / 'Bottom' keypoint (ellipse.py:403-426, :472-488): by image y
    when the two are vertically separated, by image x (direction decided by where the left-side annotations / the arc lie)
    when the cutting line runs almost horizontally in the image."""
    p1, p2 = np.asarray(pair[0], dtype=np.float64), np.asarray(pair[1], dtype=np.float64)
    y_min = min(p1[1], p2[1])
    left_right = False
    for name, poly in points.items():
        if 'left' in name.split()[:3] and name not in (line, circle):
            if any(p[1] * img_size[1] > y_min for p in poly):
                left_right = True
                break
    if circle == 'Circle left' and any((y_min - p[1] * img_size[1]) > 3 for p in points[circle]):
        left_right = True
    if circle == 'Circle right' and any((p[1] * img_size[1] - y_min) > 3 for p in points[circle]):
        left_right = True
    dx, dy = abs(p1[0] - p2[0]), abs(p1[1] - p2[1])
    if dy < 1.0 or dx / dy > 10:
        bottom, top = (p2, p1) if p1[0] < p2[0] else (p1, p2)
        if not left_right:
            bottom, top = top, bottom
    else:
        bottom, top = (p2, p1) if p1[1] < p2[1] else (p1, p2)
    return bottom if side == 'Bottom' else top


# ---- homography --------------------------------------------------------------------------------------------------
def _dlt(src: np.ndarray, dst: np.ndarray) -> Optional[np.ndarray]:
    def norm(p):
        m = p.mean(axis=0)
        s = np.sqrt(2.0) / max(np.sqrt(((p - m) ** 2).sum(axis=1)).mean(), 1e-12)
        return np.array([[s, 0, -s * m[0]], [0, s, -s * m[1]], [0, 0, 1.0]])
    Ts, Td = norm(src), norm(dst)
    s = (np.c_[src, np.ones(len(src))] @ Ts.T)[:, :2]
    d = (np.c_[dst, np.ones(len(dst))] @ Td.T)[:, :2]
    A = []
    for (x, y), (u, v) in zip(s, d):
        A.append([-x, -y, -1, 0, 0, 0, u * x, u * y, u])
        A.append([0, 0, 0, -x, -y, -1, v * x, v * y, v])
    try:
        _, _, vt = np.linalg.svd(np.asarray(A))
    except np.linalg.LinAlgError:
        return None
    H = np.linalg.inv(Td) @ vt[-1].reshape(3, 3) @ Ts
    return H / H[2, 2] if abs(H[2, 2]) > 1e-300 else None


def homography_ransac(src: np.ndarray, dst: np.ndarray, threshold: float = 5.0, iters: int = 200) -> Optional[np.ndarray]:
    """src -> dst homography from >= 4 correspondences: 4-point normalised-DLT hypotheses over a fixed pseudo-random sample
    sequence, inliers at `threshold` pixels of transfer error, final DLT on the inliers of the best hypothesis."""
    src, dst = np.asarray(src, dtype=np.float64), np.asarray(dst, dtype=np.float64)
    n = len(src)
    if n < 4:
        return None
    rng = np.random.Generator(np.random.PCG64(12345))

    def err(H):
        p = np.c_[src, np.ones(n)] @ H.T
        with np.errstate(divide='ignore', invalid='ignore'):
            q = p[:, :2] / p[:, 2:3]
        return np.sqrt(((q - dst) ** 2).sum(axis=1))
    best, best_in = None, None
    for _ in range(iters if n > 4 else 1):
        idx = rng.choice(n, 4, replace=False) if n > 4 else np.arange(4)
        H = _dlt(src[idx], dst[idx])
        if H is None:
            continue
        inl = err(H) < threshold
        if best_in is None or inl.sum() > best_in.sum():
            best, best_in = H, inl
    if best is None or best_in.sum() < 4:
        return best
    H = _dlt(src[best_in], dst[best_in])
    return H if H is not None else best


# ---- the labels ----------------------------------------------------------------------------------------------------
def add_conic_points(points: Dict[str, List], labels: Dict[int, Optional[Tuple[float, float]]], img_size=(960, 540)):
    size = np.asarray(img_size, dtype=np.float64)
    for conic, wanted in CONIC_POINTS.items():
        q = fit_ellipse(np.asarray(points[conic], dtype=np.float64) * size) if conic in points and len(points[conic]) > 4 else None
        if q is None:
            continue
        for name, kind, arg, sel in wanted:
            kid = _ID[name]
            if kind == 'tangent':
                ref = labels.get(_ID[arg])
                if ref is not None:
                    tp = tangent_points(q, ref)
                    # no real tangent (reference point inside the fitted ellipse: a broken annotation): the reference's closed
                    # form turns into NaN coordinates, which neither pass the image test nor get replaced by the homography
                    labels[kid] = (float(tp[sel][0]), float(tp[sel][1])) if tp is not None else (float('nan'), float('nan'))
            elif arg in points and len(points[arg]) > 1:
                pair = conic_cross_line(q, np.asarray(points[arg], dtype=np.float64) * size)
                if pair is not None:
                    p = pick_side(pair, points, img_size, conic, arg, sel)
                    labels[kid] = (float(p[0]), float(p[1]))
    # circle points still missing: through the pitch -> image homography of what is known (ground-plane points only)
    known = [i for i, p in labels.items() if p is not None and i not in NOT_ON_PLANE and p[0] == p[0]]
    mask: List[int] = []
    H = None
    if len(known) > 3:
        world = np.array([PITCH_POINTS[INTERSECTON_TO_PITCH_POINTS[i]][:2] for i in known], dtype=np.float32)
        image = np.array([labels[i] for i in known], dtype=np.float32)
        H = homography_ransac(world, image, 5.0)
    for i in INTERSECTON_TO_PITCH_POINTS:
        if i > 29 and labels.get(i) is None:
            if H is not None:
                w = PITCH_POINTS[INTERSECTON_TO_PITCH_POINTS[i]]
                h = H @ np.array([w[0], w[1], 1.0])
                labels[i] = (float(h[0] / h[2]), float(h[1] / h[2]))
            else:
                mask.append(i)
    for i in INTERSECTON_TO_PITCH_POINTS:
        labels.setdefault(i, None)
    return labels, mask


def get_intersections(points: Dict[str, List[Tuple[float, float]]], img_size: Tuple[int, int] = (960, 540),
                      within_image: bool = True, margin: float = 0.0):
    """{annotation class: [(x, y) normalised]} -> ({keypoint id: (x_px, y_px) or None}, [ids of circle points not produced])."""
    size = np.asarray(img_size, dtype=np.float64)
    labels: Dict[int, Optional[Tuple[float, float]]] = {}
    for i, (n1, n2) in LINE_INTERSECTIONS.items():
        labels[i] = None
        if n1 in points and n2 in points and len(points[n1]) > 1 and len(points[n2]) > 1:
            labels[i] = inside(line_intersection(np.asarray(points[n1], dtype=np.float64) * size,
                                                 np.asarray(points[n2], dtype=np.float64) * size), img_size, within_image, margin)
    labels, mask = add_conic_points(points, labels, img_size)
    return {i: inside(p, img_size, True, margin) for i, p in labels.items()}, mask
