"""Batched camera evaluation on the GPU: the SoccerNet calibration accuracy@t metric.

Host side of sncal_evaluate_cameras (csrc/evaluate.hip).  Mirrors
  SoccerPitch.line_extremities / sample_field_points / symetric_classes  /root/reference/baseline/soccerpitch.py:46-75, 264-318, 420-510
  scale_points / mirror_labels                                          /root/reference/baseline/evaluate_extremities.py:24-34, 119-134
  the per-frame loop and the aggregation of evaluate_camera.py:265-345  (completeness x mean accuracy = final score)
The pitch samples are computed HERE with numpy (not in the kernel): their cos / sin / arctan2 must be numpy's for the
table to equal the reference's bit for bit (tests/golden/evaluator_batch.npz pins it).
"""
import ctypes
from typing import Dict, List, Optional, Sequence

import numpy as np

from . import _lib
from .pitch import PITCH_POINTS, CENTER_CIRCLE_RADIUS

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
CLASSES: List[str] = ['Circle central'] + [e[0] for e in LINE_EXTREMITIES]

_PAIRS = [('Side line top', 'Side line bottom'), ('Side line left', 'Side line right'),
          ('Big rect. left top', 'Big rect. right bottom'), ('Big rect. left bottom', 'Big rect. right top'),
          ('Big rect. left main', 'Big rect. right main'), ('Small rect. left top', 'Small rect. right bottom'),
          ('Small rect. left bottom', 'Small rect. right top'), ('Small rect. left main', 'Small rect. right main'),
          ('Circle left', 'Circle right'), ('Goal left crossbar', 'Goal right crossbar'),
          ('Goal left post left ', 'Goal right post left'), ('Goal left post right', 'Goal right post right')]
SYMMETRIC: Dict[str, str] = {k: k for k in ('Middle line', 'Circle central', 'Goal unknown', 'Line unknown')}
for _a, _b in _PAIRS:
    SYMMETRIC[_a], SYMMETRIC[_b] = _b, _a


def sample_field_points(dist: float = 0.1, dist_circles: float = 0.2) -> Dict[str, List[np.ndarray]]:
    """soccerpitch.py:420-510."""
    P, R = PITCH_POINTS, CENTER_CIRCLE_RADIUS
    out = {}

    def arc(center, from_a, to_a, closed):
        poly = [np.array((center[0] + np.cos(from_a) * R, center[1] + np.sin(from_a) * R, 0.))]
        nb = int(R * (to_a - from_a) / dist_circles)
        dangle = dist_circles / R
        for i in range(1, nb if closed else nb + 1):
            a = from_a + i * dangle
            poly.append(np.array((center[0] + np.cos(a) * R, center[1] + np.sin(a) * R, 0)))
        if not closed:
            poly.append(np.array((center[0] + np.cos(to_a) * R, center[1] + np.sin(to_a) * R, 0.)))
        return poly
    out['Circle central'] = arc(P['CENTER_MARK'], 0.0, 2 * np.pi, True)
    for key, k0, k1 in LINE_EXTREMITIES:
        if key == 'Circle right':
            c, top, bottom = P['R_PENALTY_MARK'], P[k0], P[k1]
            to_a = np.arctan2(top[1] - c[1], top[0] - c[0]) + 2 * np.pi
            from_a = np.arctan2(bottom[1] - c[1], bottom[0] - c[0]) + 2 * np.pi
            out[key] = arc(c, from_a, to_a + 2 * np.pi if to_a < from_a else to_a, False)
        elif key == 'Circle left':
            c, top, bottom = P['L_PENALTY_MARK'], P[k0], P[k1]
            from_a = np.arctan2(top[1] - c[1], top[0] - c[0]) + 2 * np.pi
            to_a = np.arctan2(bottom[1] - c[1], bottom[0] - c[0]) + 2 * np.pi
            out[key] = arc(c, from_a, to_a + 2 * np.pi if to_a < from_a else to_a, False)
        else:
            start, end = np.array(P[k0], dtype=float), np.array(P[k1], dtype=float)
            poly = [start]
            nb = int(np.sqrt(np.sum(np.square(start - end))) / dist - 1)
            v = end - start
            v /= np.linalg.norm(v)
            prev = start
            for _ in range(nb):
                prev = prev + dist * v
                poly.append(prev)
            poly.append(end)
            out[key] = poly
    return out


def field_table(dist: float = 0.9, dist_circles: float = 0.2):
    """(points (N,3) float64, class_start (len(CLASSES)+1,) int32), CLASSES order."""
    s = sample_field_points(dist, dist_circles)
    pts, start = [], [0]
    for c in CLASSES:
        pts += s[c]
        start.append(len(pts))
    return np.array(pts, dtype=np.float64), np.array(start, dtype=np.int32)


def scale_points(points_dict, s_width, s_height):
    """evaluate_extremities.py:119-134: normalised annotations -> pixels; empty classes dropped."""
    out = {}
    for cls, pts in points_dict.items():
        scaled = [{'x': p['x'] * (s_width - 1), 'y': p['y'] * (s_height - 1)} for p in pts]
        if scaled:
            out[cls] = scaled
    return out


class CameraEvaluator:
    """evaluate(records, annotations) -> per-frame confusion / accuracy arrays and the benchmark summary.

    records      (B, sizeof(sncal_camera)) uint8 cuda tensor from CameraCreator.solve_device (status 0 = missed frame)
    annotations  per frame {class name: [{'x':..,'y':..} or (x, y), ...]} in PIXELS (scale_points does the SoccerNet
                 normalised -> pixel step)
    """

    def __init__(self, device, width: int = 960, height: int = 540, threshold: float = 5.0, sampling_factor: float = 0.9):
        import torch
        self.device = torch.device(device)
        self.width, self.height, self.threshold = int(width), int(height), float(threshold)
        pts, start = field_table(sampling_factor)
        self._field = torch.from_numpy(pts).to(self.device)
        self._start = torch.from_numpy(start).to(self.device)
        idx = {c: i for i, c in enumerate(CLASSES)}
        self._mirror = torch.tensor([idx[SYMMETRIC[c]] for c in CLASSES], dtype=torch.int32, device=self.device)
        self._idx = idx

    def pack(self, annotations: Sequence[Dict[str, list]]):
        B, C = len(annotations), len(CLASSES)
        max_gt = max([1] + [len(v) for a in annotations for k, v in a.items() if k in self._idx])
        gt = np.zeros((B, C, max_gt, 2), dtype=np.float64)
        cnt = np.zeros((B, C), dtype=np.int32)
        extra = np.zeros((B,), dtype=np.int32)
        for b, ann in enumerate(annotations):
            for cls, pts in ann.items():
                if cls not in self._idx:
                    extra[b] += 1
                    continue
                c = self._idx[cls]
                cnt[b, c] = len(pts)
                for k, p in enumerate(pts):
                    gt[b, c, k] = (p['x'], p['y']) if isinstance(p, dict) else (p[0], p[1])
        return gt, cnt, extra, max_gt

    def evaluate(self, records, annotations: Sequence[Dict[str, list]], detail: bool = False):
        """-> out (B,12) fp32 [confusion plain, confusion mirrored, accuracy plain, accuracy mirrored, chosen pass,
        evaluated]; with detail=True -> (out, err (B,2,C,max_gt) fp64, class_conf (B,2,C,4) int32), see
        sncal_evaluate_cameras_detail and class_report()."""
        import torch
        _lib.require_device(records, torch.uint8, 'records')
        B = records.shape[0]
        if len(annotations) != B or records.shape[1] != ctypes.sizeof(_lib.Camera):
            raise _lib.SncalError('records must be (B, sizeof(sncal_camera)) with one annotation dict per frame')
        gt, cnt, extra, max_gt = self.pack(annotations)
        d_gt, d_cnt, d_extra = (torch.from_numpy(a).to(self.device) for a in (gt, cnt, extra))
        out = torch.empty((B, 12), dtype=torch.float32, device=self.device)
        C = len(CLASSES)
        err = torch.empty((B, 2, C, max_gt), dtype=torch.float64, device=self.device) if detail else None
        cls = torch.empty((B, 2, C, 4), dtype=torch.int32, device=self.device) if detail else None
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().sncal_evaluate_cameras_detail(
                records.data_ptr(), B, self._field.data_ptr(), self._start.data_ptr(), self._mirror.data_ptr(), C,
                d_gt.data_ptr(), d_cnt.data_ptr(), d_extra.data_ptr(), max_gt, self.threshold, self.width, self.height,
                out.data_ptr(), err.data_ptr() if detail else None, cls.data_ptr() if detail else None,
                _lib.current_stream_ptr()), 'sncal_evaluate_cameras_detail')
        return (out, err, cls) if detail else out

    @staticmethod
    def frame_detail(out, err, cls, annotations, b: int, which: int = 0):
        """The `per_class_confusion` / `dict_errors` dictionaries evaluate_camera_prediction (evaluate_camera.py:172-226)
        returns for frame b: which = 1 plain labels, 2 mirrored labels, 0 the pass evaluate_camera.py:303-311 keeps."""
        o = out[b].detach().cpu().numpy() if hasattr(out, 'detach') else np.asarray(out[b])
        p = (int(o[10]) if which == 0 else which) - 1
        e = err[b, p].detach().cpu().numpy() if hasattr(err, 'detach') else np.asarray(err[b, p])
        q = cls[b, p].detach().cpu().numpy() if hasattr(cls, 'detach') else np.asarray(cls[b, p])
        per_class, errors = {}, {}
        for c, name in enumerate(CLASSES):
            below, beyond, missed, fp_flag = (int(v) for v in q[c])
            if fp_flag:
                per_class[name] = np.array([[0., 9. if 'Circle' in name else 2.], [0., 0.]])
            elif missed:
                per_class[name] = np.array([[0., 0.], [float(missed), 0.]])
            elif below or beyond:
                per_class[name] = np.array([[float(below), float(beyond)], [0., 0.]])
                errors[name] = [float(v) for v in e[c, :below + beyond]]
        for name, pts in annotations[b].items():                          # annotated classes the pitch model does not have
            if name not in CLASSES and len(pts):
                per_class[name] = np.array([[0., 0.], [float(len(pts)), 0.]])
        return per_class, errors

    @classmethod
    def class_report(cls_, out, err, cls, annotations, bins: int = 30, hist_range=(0.0, 60.0)):
        """evaluate_camera.py:335-373 over a batch: accumulated per-class confusion -> accuracy / precision / recall per
        class, and the per-class reprojection-error histograms the script plots (counts, bin edges)."""
        o = out.detach().cpu().numpy() if hasattr(out, 'detach') else np.asarray(out)
        err = err.detach().cpu().numpy() if hasattr(err, 'detach') else np.asarray(err)
        cls = cls.detach().cpu().numpy() if hasattr(cls, 'detach') else np.asarray(cls)
        conf, errors = {}, {}
        for b in range(len(o)):
            if o[b, 11] <= 0:
                continue
            pc, er = cls_.frame_detail(o, err, cls, annotations, b)
            for k, m in pc.items():
                conf[k] = conf.get(k, 0) + m
            for k, v in er.items():
                errors.setdefault(k, []).extend(v)
        report = {}
        with np.errstate(divide='ignore', invalid='ignore'):
            for k, m in conf.items():
                report[k] = {'confusion': m, 'accuracy': float(m[0, 0] / m.sum()),
                             'recall': float(np.float64(m[0, 0]) / (m[0, 0] + m[1, 0])),
                             'precision': float(np.float64(m[0, 0]) / (m[0, 0] + m[0, 1]))}
        hist = {k: np.histogram(np.asarray(v)[np.isfinite(v)], bins=bins, range=hist_range) for k, v in errors.items()}
        return report, errors, hist

    @staticmethod
    def summarize(out) -> Dict[str, float]:
        """evaluate_camera.py:321-345: completeness, mean accuracy over evaluated frames, final score, precision/recall."""
        o = out.detach().cpu().numpy() if hasattr(out, 'detach') else np.asarray(out)
        done = o[:, 11] > 0
        total = len(o)
        chosen = np.where((o[:, 10] == 1)[:, None], o[:, 0:4], o[:, 4:8])[done]
        acc = np.where(o[:, 10] == 1, o[:, 8], o[:, 9])[done]
        prec = [c[0] / (c[0] + c[1]) for c in chosen if c[0] + c[1] > 0]
        rec = [c[0] / (c[0] + c[2]) for c in chosen if c[0] + c[2] > 0]
        completeness = float(done.sum()) / total if total else 0.0
        macc = float(np.mean(acc)) if len(acc) else 0.0
        return {'completeness': completeness, 'accuracy': macc, 'final_score': completeness * macc,
                'precision': float(np.mean(prec)) if prec else 0.0, 'recall': float(np.mean(rec)) if rec else 0.0}
