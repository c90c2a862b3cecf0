"""Oracle: line-model post-processing and the line -> keypoint join.  TEST INFRASTRUCTURE ONLY.

L3  /root/reference/src/utils/export_line_result.py:51-82 (calculate_slope_intercept),
    :85-131 (get_line_data);  LINE_CLS /root/reference/src/datatools/line.py:35-57
L4  /root/reference/src/models/hrnet/prediction.py:105-124 (ingestion), :643-653
    (line_eq_intersection);  LINE_INTERSECTIONS /root/reference/src/datatools/intersections.py:13-44
Pinned by tests/golden/lines.json (captured from the imported reference functions).
"""
from typing import Dict, List, Optional, Tuple

import numpy as np

LINE_CLS: List[str] = [
    'Goal left post left ', 'Goal right post right', 'Middle line', 'Small rect. right top',
    'Side line bottom', 'Goal right post left', 'Big rect. right main', 'Goal left crossbar',
    'Small rect. left bottom', 'Side line left', 'Big rect. right top', 'Small rect. left top',
    'Side line right', 'Big rect. left top', 'Goal left post right', 'Small rect. right bottom',
    'Side line top', 'Goal right crossbar', 'Small rect. left main', 'Big rect. left main',
    'Big rect. right bottom', 'Small rect. right main', 'Big rect. left bottom']

# keypoint id -> the two line classes whose intersection defines it (ids 0..29)
LINE_INTERSECTIONS: Dict[int, Tuple[str, str]] = {
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
}


def slope_intercept(p1, p2, delta: float = 0.00001):
    """export_line_result.py:51-82 with the scalar promotion of the reference's pinned numpy 1.24.2: float32
    coordinates stay float32 through the differences, `+ delta` (a python float) promotes to float64."""
    if tuple(p1) == tuple(p2):
        return None, None
    x1, y1 = p1
    x2, y2 = p2
    if isinstance(x1, np.float32):
        dy, dx = np.float32(y2) - np.float32(y1), np.float32(x2) - np.float32(x1)
        slope = float(dy) / (float(dx) + delta)
        return slope, float(y1) - slope * float(x1)
    slope = (y2 - y1) / (x2 - x1 + delta)
    return slope, y1 - slope * x1


def get_line_data(heat_loc: np.ndarray, scale=4, prob_thre: float = 0.2):
    """export_line_result.py:85-131.  heat_loc (1,K,2,3) [x,y,p]."""
    _, ks, nh, _ = heat_loc.shape
    lines, points = {}, {}
    for k in range(ks):
        valid = []
        for n in range(nh):
            x, y, p = heat_loc[0, k, n]
            if p >= prob_thre:
                valid.append((x * scale, y * scale, p))
        points[LINE_CLS[k]] = valid
        if len(valid) >= 2:
            lines[LINE_CLS[k]] = slope_intercept(valid[0][:2], valid[1][:2])
    return lines, points


def line_eq_intersection(l1, l2) -> Optional[Tuple[float, float]]:
    """prediction.py:643-653."""
    k1, b1 = l1
    k2, b2 = l2
    if abs(k1 - k2) > 1e-4:
        x = (b2 - b1) / (k1 - k2)
        return (x, k1 * x + b1)
    return None


def lines_to_keypoints(pred: Dict[str, Tuple[float, float]]) -> Dict[int, Tuple[float, float]]:
    """prediction.py:110-124 for one image's {line name: (k, b)} dict."""
    pred = dict(pred)
    if 'Goal left post left' in pred:  # key without the trailing blank (:113-115)
        pred['Goal left post left '] = pred.pop('Goal left post left')
    pts = {}
    for idx, (a, b) in LINE_INTERSECTIONS.items():
        if a in pred and b in pred:
            ip = line_eq_intersection(pred[a], pred[b])
            if ip is not None:
                pts[idx] = ip
    return pts


def keypoints_array(peaks: np.ndarray, scale=4, prob_thre: float = 0.2) -> np.ndarray:
    """L3 + L4 for a batch: peaks (B,23,2,3) float32 -> (B,30,3) float32 rows [x, y, valid] (the d_line_pts of the
    solver).  A (None, None) line (two identical peaks; the reference would raise on it at prediction.py:650) counts
    as missing."""
    out = np.zeros((peaks.shape[0], 30, 3), dtype=np.float32)
    for b in range(peaks.shape[0]):
        lines, _ = get_line_data(peaks[b:b + 1], scale=np.float32(scale) if isinstance(scale, float) else scale, prob_thre=prob_thre)
        lines = {k: v for k, v in lines.items() if v[0] is not None}
        for i, (x, y) in lines_to_keypoints(lines).items():
            out[b, i] = (x, y, 1.0)
    return out
