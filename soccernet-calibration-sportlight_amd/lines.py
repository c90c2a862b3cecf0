"""Line-model post-processing and the line -> keypoint join (host logic, python floats like the reference).

LINE_CLS            <-> /root/reference/src/datatools/line.py:35-57
LINE_INTERSECTIONS  <-> /root/reference/src/datatools/intersections.py:13-44
calculate_slope_intercept / get_line_data <-> /root/reference/src/utils/export_line_result.py:51-131
line_eq_intersection / lines_to_keypoints <-> /root/reference/src/models/hrnet/prediction.py:105-124, 643-653
The pickle schema {img: {'lines': [ {name: (k, b)} ], 'points': [ {name: [(x,y,p)]} ]}} is the reference's
(export_line_result.py:188, 200-201; consumed at prediction.py:107-124).
"""
from typing import Dict, List, Optional, Tuple

import numpy as np

LINE_CLS: Dict[int, str] = dict(enumerate([
    'Goal left post left ', 'Goal right post right', 'Middle line', 'Small rect. right top', 'Side line bottom',
    'Goal right post left', 'Big rect. right main', 'Goal left crossbar', 'Small rect. left bottom',
    'Side line left', 'Big rect. right top', 'Small rect. left top', 'Side line right', 'Big rect. left top',
    'Goal left post right', 'Small rect. right bottom', 'Side line top', 'Goal right crossbar',
    'Small rect. left main', 'Big rect. left main', 'Big rect. right bottom', 'Small rect. right main',
    'Big rect. left bottom']))

LINE_INTERSECTIONS: Dict[int, Tuple[str, str]] = {}
LINE_INTERSECTIONS.update({
    0: ('Goal left crossbar', 'Goal left post left '), 1: ('Goal left crossbar', 'Goal left post right'),
    2: ('Side line left', 'Goal left post left '), 3: ('Side line left', 'Goal left post right'),
    4: ('Small rect. left main', 'Small rect. left bottom'), 5: ('Small rect. left main', 'Small rect. left top'),
    6: ('Side line left', 'Small rect. left bottom'), 7: ('Side line left', 'Small rect. left top'),
    8: ('Big rect. left main', 'Big rect. left bottom'), 9: ('Big rect. left main', 'Big rect. left top'),
    10: ('Side line left', 'Big rect. left bottom'), 11: ('Side line left', 'Big rect. left top'),
    12: ('Side line left', 'Side line bottom'), 13: ('Side line left', 'Side line top'),
    14: ('Middle line', 'Side line bottom'), 15: ('Middle line', 'Side line top'),
    16: ('Big rect. right main', 'Big rect. right bottom'), 17: ('Big rect. right main', 'Big rect. right top'),
    18: ('Side line right', 'Big rect. right bottom'), 19: ('Side line right', 'Big rect. right top'),
    20: ('Small rect. right main', 'Small rect. right bottom'), 21: ('Small rect. right main', 'Small rect. right top'),
    22: ('Side line right', 'Small rect. right bottom'), 23: ('Side line right', 'Small rect. right top'),
    24: ('Goal right crossbar', 'Goal right post left'), 25: ('Goal right crossbar', 'Goal right post right'),
    26: ('Side line right', 'Goal right post left'), 27: ('Side line right', 'Goal right post right'),
    28: ('Side line right', 'Side line bottom'), 29: ('Side line right', 'Side line top'),
})


def calculate_slope_intercept(point1, point2, delta: float = 0.00001):
    """(slope, intercept) of the line through two points, (None, None) for coincident ones; `delta` keeps vertical lines finite.
    float32 numpy coordinates follow the reference's pinned numpy 1.24.2 promotion (rise and run formed in float32, float64
    from `+ delta` on) whatever numpy is installed; python floats are plain float64 arithmetic."""
    (xa, ya), (xb, yb) = point1, point2
    if xa == xb and ya == yb:
        return None, None
    if isinstance(xa, np.float32):
        rise, run = float(np.float32(yb) - np.float32(ya)), float(np.float32(xb) - np.float32(xa))
        xa, ya = float(xa), float(ya)
    else:
        rise, run = yb - ya, xb - xa
    k = rise / (run + delta)
    return k, ya - k * xa


def get_line_data(heat_loc, scale=4, prob_thre: float = 0.2):
    """heat_loc (1,K,2,3) rows [x, y, p] (numpy or torch) -> ({name: (k, b)}, {name: [(x,y,p)]}): per line class the peaks at or
    above the threshold in image units, and the line through the first two of them."""
    peaks = heat_loc.cpu().numpy() if hasattr(heat_loc, 'cpu') else np.asarray(heat_loc)
    lines, points = {}, {}
    for cls, rows in enumerate(peaks[0]):
        kept = [(x * scale, y * scale, p) for x, y, p in rows if p >= prob_thre]
        points[LINE_CLS[cls]] = kept
        if len(kept) > 1:
            lines[LINE_CLS[cls]] = calculate_slope_intercept(kept[0][:2], kept[1][:2])
    return lines, points


def line_eq_intersection(line1, line2) -> Optional[Tuple[float, float]]:
    """Intersection of y = k x + b lines; None for (nearly) parallel ones (slopes within 1e-4, NaN slopes included)."""
    (ka, ba), (kb, bb) = line1, line2
    dk = ka - kb
    if not abs(dk) > 1e-4:
        return None
    x = (bb - ba) / dk
    return (x, ka * x + ba)


def lines_to_keypoints(pred: Dict[str, Tuple[float, float]]) -> Dict[int, Tuple[float, float]]:
    """One image's {line name: (slope, intercept)} -> {keypoint id 0..29: (x, y)} (prediction.py:110-124)."""
    pred = dict(pred)
    if 'Goal left post left' in pred:
        pred['Goal left post left '] = pred.pop('Goal left post left')
    points = {}
    for idx, pair in LINE_INTERSECTIONS.items():
        if pair[0] in pred and pair[1] in pred:
            ip = line_eq_intersection(pred[pair[0]], pred[pair[1]])
            if ip is not None:
                points[idx] = ip
    return points


def keypoints_to_array(points: Dict[int, Tuple[float, float]]) -> np.ndarray:
    """{id: (x,y)} -> (30,3) float32 rows [x, y, valid] for sncal_calibrate's d_line_pts."""
    a = np.zeros((30, 3), dtype=np.float32)
    for i, (x, y) in points.items():
        a[i] = (x, y, 1.0)
    return a


def lines_to_points_device(peaks, scale: float = 4.0, prob_thre: float = 0.2):
    """L3 + L4 on the GPU for a batch: peaks (B,23,2,3) float32 cuda tensor (line decode in heatmap units) ->
    (B,30,3) float32 cuda tensor [x, y, valid], the `d_line_pts` argument of CameraCreator.solve_device."""
    import torch
    from . import _lib
    _lib.require_device(peaks, torch.float32, 'peaks')
    if tuple(peaks.shape[1:]) != (23, 2, 3):
        raise _lib.SncalError('peaks must be (B,23,2,3)')
    out = torch.empty((peaks.shape[0], 30, 3), dtype=torch.float32, device=peaks.device)
    with torch.cuda.device(peaks.device):
        _lib.check(_lib.lib().sncal_lines_to_points(peaks.data_ptr(), peaks.shape[0], float(scale), float(prob_thre),
                                                    out.data_ptr(), _lib.current_stream_ptr()), 'sncal_lines_to_points')
    return out
