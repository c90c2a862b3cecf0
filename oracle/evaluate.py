"""Oracle: camera evaluation metric (SoccerNet calibration Acc@t).  TEST INFRASTRUCTURE ONLY.

Follows /root/reference/baseline/soccerpitch.py:264-318 (line_extremities), :420-510 (sample_field_points), :46-75
(symetric_classes); /root/reference/baseline/evaluate_camera.py:14-105 (get_polylines), :108-157
(distance_to_polyline), :160-229 (evaluate_camera_prediction), :293-320 (mirrored evaluation, accuracy choice);
/root/reference/baseline/evaluate_extremities.py:12-21 (distance), :24-34 (mirror_labels), :119-134 (scale_points).
Pinned by tests/golden/evaluator.json and tests/golden/evaluator_batch.npz (captured from the imported reference).
"""
import numpy as np

from . import camera_math as cm
from .pitch import pitch_points

# keypoint ids of the named pitch points used below (ellipse.py:99-157, INTERSECTON_TO_PITCH_POINTS)
_ID = {'L_GOAL_TL_POST': 0, 'L_GOAL_TR_POST': 1, 'L_GOAL_BL_POST': 2, 'L_GOAL_BR_POST': 3, 'L_GOAL_AREA_BR_CORNER': 4,
       'L_GOAL_AREA_TR_CORNER': 5, 'L_GOAL_AREA_BL_CORNER': 6, 'L_GOAL_AREA_TL_CORNER': 7, 'L_PENALTY_AREA_BR_CORNER': 8,
       'L_PENALTY_AREA_TR_CORNER': 9, 'L_PENALTY_AREA_BL_CORNER': 10, 'L_PENALTY_AREA_TL_CORNER': 11, 'BL_PITCH_CORNER': 12,
       'TL_PITCH_CORNER': 13, 'B_TOUCH_AND_HALFWAY_LINES_INTERSECTION': 14, 'T_TOUCH_AND_HALFWAY_LINES_INTERSECTION': 15,
       'R_PENALTY_AREA_BL_CORNER': 16, 'R_PENALTY_AREA_TL_CORNER': 17, 'R_PENALTY_AREA_BR_CORNER': 18,
       'R_PENALTY_AREA_TR_CORNER': 19, 'R_GOAL_AREA_BL_CORNER': 20, 'R_GOAL_AREA_TL_CORNER': 21, 'R_GOAL_AREA_BR_CORNER': 22,
       'R_GOAL_AREA_TR_CORNER': 23, 'R_GOAL_TL_POST': 24, 'R_GOAL_TR_POST': 25, 'R_GOAL_BL_POST': 26, 'R_GOAL_BR_POST': 27,
       'BR_PITCH_CORNER': 28, 'TR_PITCH_CORNER': 29, 'CENTER_MARK': 42, 'BL_16M_LINE_AND_PENALTY_ARC_INTERSECTION': 44,
       'TL_16M_LINE_AND_PENALTY_ARC_INTERSECTION': 45, 'L_PENALTY_MARK': 48, 'BR_16M_LINE_AND_PENALTY_ARC_INTERSECTION': 51,
       'TR_16M_LINE_AND_PENALTY_ARC_INTERSECTION': 52, 'R_PENALTY_MARK': 55}
_PP = pitch_points()
PITCH_POINTS = {k: _PP[i].copy() for k, i in _ID.items()}

R = 9.15        # SoccerPitch.CENTER_CIRCLE_RADIUS

# soccerpitch.py:264-318, dict insertion order
LINE_EXTREMITIES = [
    ('Big rect. left bottom', 'L_PENALTY_AREA_BL_CORNER', 'L_PENALTY_AREA_BR_CORNER'),
    ('Big rect. left top', 'L_PENALTY_AREA_TL_CORNER', 'L_PENALTY_AREA_TR_CORNER'),
    ('Big rect. left main', 'L_PENALTY_AREA_TR_CORNER', 'L_PENALTY_AREA_BR_CORNER'),
    ('Big rect. right bottom', 'R_PENALTY_AREA_BL_CORNER', 'R_PENALTY_AREA_BR_CORNER'),
    ('Big rect. right top', 'R_PENALTY_AREA_TL_CORNER', 'R_PENALTY_AREA_TR_CORNER'),
    ('Big rect. right main', 'R_PENALTY_AREA_TL_CORNER', 'R_PENALTY_AREA_BL_CORNER'),
    ('Small rect. left bottom', 'L_GOAL_AREA_BL_CORNER', 'L_GOAL_AREA_BR_CORNER'),
    ('Small rect. left top', 'L_GOAL_AREA_TL_CORNER', 'L_GOAL_AREA_TR_CORNER'),
    ('Small rect. left main', 'L_GOAL_AREA_TR_CORNER', 'L_GOAL_AREA_BR_CORNER'),
    ('Small rect. right bottom', 'R_GOAL_AREA_BL_CORNER', 'R_GOAL_AREA_BR_CORNER'),
    ('Small rect. right top', 'R_GOAL_AREA_TL_CORNER', 'R_GOAL_AREA_TR_CORNER'),
    ('Small rect. right main', 'R_GOAL_AREA_TL_CORNER', 'R_GOAL_AREA_BL_CORNER'),
    ('Side line top', 'TL_PITCH_CORNER', 'TR_PITCH_CORNER'),
    ('Side line bottom', 'BL_PITCH_CORNER', 'BR_PITCH_CORNER'),
    ('Side line left', 'TL_PITCH_CORNER', 'BL_PITCH_CORNER'),
    ('Side line right', 'TR_PITCH_CORNER', 'BR_PITCH_CORNER'),
    ('Middle line', 'T_TOUCH_AND_HALFWAY_LINES_INTERSECTION', 'B_TOUCH_AND_HALFWAY_LINES_INTERSECTION'),
    ('Goal left crossbar', 'L_GOAL_TR_POST', 'L_GOAL_TL_POST'),
    ('Goal left post left ', 'L_GOAL_TL_POST', 'L_GOAL_BL_POST'),
    ('Goal left post right', 'L_GOAL_TR_POST', 'L_GOAL_BR_POST'),
    ('Goal right crossbar', 'R_GOAL_TL_POST', 'R_GOAL_TR_POST'),
    ('Goal right post left', 'R_GOAL_TL_POST', 'R_GOAL_BL_POST'),
    ('Goal right post right', 'R_GOAL_TR_POST', 'R_GOAL_BR_POST'),
    ('Circle right', 'TR_16M_LINE_AND_PENALTY_ARC_INTERSECTION', 'BR_16M_LINE_AND_PENALTY_ARC_INTERSECTION'),
    ('Circle left', 'TL_16M_LINE_AND_PENALTY_ARC_INTERSECTION', 'BL_16M_LINE_AND_PENALTY_ARC_INTERSECTION'),
]
CLASSES = ['Circle central'] + [e[0] for e in LINE_EXTREMITIES]          # iteration order of sample_field_points

SYMMETRIC = {       # soccerpitch.py:46-75
    'Side line top': 'Side line bottom', 'Side line bottom': 'Side line top', 'Side line left': 'Side line right',
    'Middle line': 'Middle line', 'Side line right': 'Side line left',
    'Big rect. left top': 'Big rect. right bottom', 'Big rect. left bottom': 'Big rect. right top',
    'Big rect. left main': 'Big rect. right main', 'Big rect. right top': 'Big rect. left bottom',
    'Big rect. right bottom': 'Big rect. left top', 'Big rect. right main': 'Big rect. left main',
    'Small rect. left top': 'Small rect. right bottom', 'Small rect. left bottom': 'Small rect. right top',
    'Small rect. left main': 'Small rect. right main', 'Small rect. right top': 'Small rect. left bottom',
    'Small rect. right bottom': 'Small rect. left top', 'Small rect. right main': 'Small rect. left main',
    'Circle left': 'Circle right', 'Circle central': 'Circle central', 'Circle right': 'Circle left',
    'Goal left crossbar': 'Goal right crossbar', 'Goal left post left ': 'Goal right post left',
    'Goal left post right': 'Goal right post right', 'Goal right crossbar': 'Goal left crossbar',
    'Goal right post left': 'Goal left post left ', 'Goal right post right': 'Goal left post right',
    'Goal unknown': 'Goal unknown', 'Line unknown': 'Line unknown',
}


def sample_field_points(dist=0.1, dist_circles=0.2):
    """soccerpitch.py:420-510: {class: [3-D points]} sampled every `dist` m (circles: `dist_circles`)."""
    P = PITCH_POINTS
    out = {}
    center = P['CENTER_MARK']
    from_a, to_a = 0.0, 2 * np.pi
    poly = [np.array((center[0] + np.cos(from_a) * R, center[1] + np.sin(from_a) * R, 0.))]
    nb = int(R * (to_a - from_a) / dist_circles)
    dangle = dist_circles / R
    for i in range(1, nb):
        a = from_a + i * dangle
        poly.append(np.array((center[0] + np.cos(a) * R, center[1] + np.sin(a) * R, 0)))
    out['Circle central'] = poly
    for key, k0, k1 in LINE_EXTREMITIES:
        if 'Circle' in key:
            if key == 'Circle right':
                top, bottom = P['TR_16M_LINE_AND_PENALTY_ARC_INTERSECTION'], P['BR_16M_LINE_AND_PENALTY_ARC_INTERSECTION']
                center = P['R_PENALTY_MARK']
                to_a = np.arctan2(top[1] - center[1], top[0] - center[0]) + 2 * np.pi
                from_a = np.arctan2(bottom[1] - center[1], bottom[0] - center[0]) + 2 * np.pi
            else:
                top, bottom = P['TL_16M_LINE_AND_PENALTY_ARC_INTERSECTION'], P['BL_16M_LINE_AND_PENALTY_ARC_INTERSECTION']
                center = P['L_PENALTY_MARK']
                from_a = np.arctan2(top[1] - center[1], top[0] - center[0]) + 2 * np.pi
                to_a = np.arctan2(bottom[1] - center[1], bottom[0] - center[0]) + 2 * np.pi
            if to_a < from_a:
                to_a += 2 * np.pi
            start = np.array((center[0] + np.cos(from_a) * R, center[1] + np.sin(from_a) * R, 0.))
            end = np.array((center[0] + np.cos(to_a) * R, center[1] + np.sin(to_a) * R, 0.))
            poly = [start]
            nb = int(R * (to_a - from_a) / dist_circles)
            dangle = dist_circles / R
            for i in range(1, nb + 1):
                a = from_a + i * dangle
                poly.append(np.array((center[0] + np.cos(a) * R, center[1] + np.sin(a) * R, 0)))
            poly.append(end)
        else:
            start, end = np.array(P[k0], dtype=float), np.array(P[k1], dtype=float)
            poly = [start]
            total = np.sqrt(np.sum(np.square(start - end)))
            nb = int(total / dist - 1)
            v = end - start
            v /= np.linalg.norm(v)
            prev = start
            for _ in range(nb):
                pt = prev + dist * v
                prev = pt
                poly.append(pt)
            poly.append(end)
        out[key] = poly
    return out


def field_table(dist=0.9, dist_circles=0.2):
    """(points (N,3) float64, class_start (len(CLASSES)+1,) int32) in CLASSES order."""
    s = sample_field_points(dist, dist_circles)
    pts, start = [], [0]
    for c in CLASSES:
        pts += s[c]
        start.append(len(pts))
    return np.array(pts, dtype=np.float64), np.array(start, dtype=np.int32)


def get_polylines(position, rotation, fx, fy, pp, width, height, table=None):
    """evaluate_camera.py:14-105 -> {class: [(x, y)]} for the classes with a non-empty projection."""
    pts, start = table if table is not None else field_table()
    sides = [np.array([1, 0, 0]), np.array([1, 0, -width + 1]), np.array([0, 1, 0]), np.array([0, 1, -height + 1])]

    def edge_point(ext, prev):
        line = np.cross(ext, prev)
        cands, dists = [], []
        for side in sides:
            inter = np.cross(line, side)
            with np.errstate(divide='ignore', invalid='ignore'):
                inter = inter / inter[2]
            if 0 <= inter[0] < width and 0 <= inter[1] < height:
                cands.append(inter)
                dists.append(np.sqrt(np.sum(np.square(inter - ext))))
        return cands[int(np.argmin(dists))] if cands else None

    out = {}
    for ci, c in enumerate(CLASSES):
        plist, in_img, prev = [], False, np.zeros(3)
        for i in range(start[ci], start[ci + 1]):
            ext = cm.project_point(position, rotation, fx, fy, pp, pts[i])
            if ext[2] < 1e-5:
                continue
            if 0 <= ext[0] < width and 0 <= ext[1] < height:
                if not in_img and i > start[ci]:
                    e = edge_point(ext, prev)
                    if e is not None:
                        plist.append((e[0], e[1]))
                plist.append((ext[0], ext[1]))
                in_img = True
            elif in_img:
                e = edge_point(ext, prev)
                if e is not None:
                    plist.append((e[0], e[1]))
                in_img = False
            prev = ext
        if plist:
            out[c] = plist
    return out


def _dist(p, q):
    d = np.array([p[0], p[1]]) - np.array([q[0], q[1]])
    return np.sqrt(np.square(d).sum())


def distance_to_polyline(point, polyline):
    """evaluate_camera.py:108-157."""
    if 0 < len(polyline) < 2:
        return _dist(point, polyline[0])
    best = []
    p = np.array([point[0], point[1], 1])
    for i in range(len(polyline) - 1):
        a = np.array([polyline[i][0], polyline[i][1], 1])
        b = np.array([polyline[i + 1][0], polyline[i + 1][1], 1])
        line = np.cross(a, b)
        with np.errstate(divide='ignore', invalid='ignore'):
            line = line / np.sqrt(np.square(line[0]) + np.square(line[1]))
            proj = np.cross(np.cross(np.array([line[0], line[1], 0]), p), line)
            proj = proj / proj[2]
            v1, v2 = proj - a, b - a
            k = np.dot(v1, v2) / np.dot(v2, v2)
        if 0 < k < 1:
            best.append(np.sqrt(np.sum(np.square(proj - p))))
        else:
            best.append(np.min([_dist(point, polyline[i]), _dist(point, polyline[i + 1])]))
    return np.min(best)


def evaluate_camera_prediction(projected, groundtruth, threshold, detail=False):
    """evaluate_camera.py:160-229 -> 2x2 float32 confusion [[TP, FP], [FN, 0]] over classes; with detail=True also the
    per-class point confusions ({class: 2x2}) and the per-class reprojection errors ({class: [distance per point]})."""
    conf = np.zeros((2, 2), dtype=np.float32)
    per_class, errors = {}, {}
    det, gt = set(projected), set(groundtruth)
    for c in det - gt:
        per_class[c] = np.array([[0., 2. if "Circle" not in c else 9.], [0., 0.]])
        conf[0, 1] += 1
    for c in gt - det:
        per_class[c] = np.array([[0., 0.], [float(len(groundtruth[c])), 0.]])
        conf[1, 0] += 1
    for c in det & gt:
        ok = True
        per_class[c] = np.zeros((2, 2))
        for point in groundtruth[c]:
            d = distance_to_polyline(point, projected[c])
            if d < threshold:
                per_class[c][0, 0] += 1
            else:
                per_class[c][0, 1] += 1
                ok = False
            errors.setdefault(c, []).append(d)
        if ok:
            conf[0, 0] += 1
        else:
            conf[0, 1] += 1
    return (conf, per_class, errors) if detail else conf


def mirror_labels(d):
    return {SYMMETRIC[k]: v for k, v in d.items()}


def evaluate_frame(position, rotation, fx, fy, pp, groundtruth, threshold, width=960, height=540, table=None):
    """evaluate_camera.py:293-320 for one frame: (confusion, accuracy) of the better of plain / mirrored labels,
    plus both confusions."""
    poly = get_polylines(position, rotation, fx, fy, pp, width, height, table)
    c1 = evaluate_camera_prediction(poly, groundtruth, threshold)
    c2 = evaluate_camera_prediction(poly, mirror_labels(groundtruth), threshold)
    a1 = c1[0, 0] / c1.sum() if c1.sum() > 0 else 0.
    a2 = c2[0, 0] / c2.sum() if c2.sum() > 0 else 0.
    return (c1, a1, c1, c2) if a1 > a2 else (c2, a2, c1, c2)
