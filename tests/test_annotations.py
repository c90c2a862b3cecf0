"""CPU: annotation geometry (N4, second half) against the capture of the imported reference, tests/golden/annotations.json
(tools/make_golden.py gen_annotations: src/datatools/intersections.get_intersections on synthetic annotations, its two
third-party calls -- lsq-ellipse's fit, cv2.findHomography -- replaced by this build's routines)."""
import json
import os

import numpy as np

import sncal_amd
from sncal_amd import annotations as an


def test_labels_match_the_reference_capture(gold_dir):
    cases = json.load(open(os.path.join(gold_dir, 'annotations.json')))
    assert len(cases) >= 20
    n_line = n_circle = n_masked = 0
    for ci, c in enumerate(cases):
        pts = {k: [tuple(p) for p in v] for k, v in c['points'].items()}
        labels, mask = an.get_intersections(pts)
        ref = np.array(c['labels'], dtype=np.float64)
        assert sorted(mask) == c['mask'], ci
        for i in range(57):
            if np.isnan(ref[i, 0]):
                assert labels[i] is None, (ci, i, labels[i])
            else:
                assert labels[i] is not None and abs(labels[i][0] - ref[i, 0]) < 1e-6 and abs(labels[i][1] - ref[i, 1]) < 1e-6, (ci, i)
                n_line += i < 30
                n_circle += i >= 30
        n_masked += len(mask)
    assert n_line > 100 and n_circle > 100          # both halves of the label set are exercised


def test_geometry_primitives():
    # ellipse through points of a known rotated ellipse: the conic equation vanishes on a dense resample
    t = np.linspace(0, 2 * np.pi, 40, endpoint=False)
    c, s = np.cos(0.4), np.sin(0.4)
    x, y = 300 + 120 * np.cos(t) * c - 40 * np.sin(t) * s, 200 + 120 * np.cos(t) * s + 40 * np.sin(t) * c
    q = an.fit_ellipse(np.c_[x, y])
    a, b, cc, d, e, f = q
    assert 4 * a * cc - b * b > 0
    assert np.abs(a * x * x + b * x * y + cc * y * y + d * x + e * y + f).max() < 1e-6 * np.abs(q).max() * 1e5
    # tangents from an external point: each tangency point lies on the conic and the chord to it touches (discriminant 0)
    P = (600.0, 50.0)
    t0, t1 = an.tangent_points(q, P)
    for T in (t0, t1):
        assert abs(a * T[0] ** 2 + b * T[0] * T[1] + cc * T[1] ** 2 + d * T[0] + e * T[1] + f) < 1e-6 * np.abs(q).max() * 1e5
        pts = an.conic_line_points(q, np.array([P[0], P[1], 1.0]), np.array([T[0] - P[0], T[1] - P[1], 0.0]))
        assert len(pts) == 2 and np.hypot(*(pts[0] - pts[1])) < 1e-3           # double root
    assert an.tangent_points(q, (300.0, 200.0)) is None                       # inside: no real tangent
    # two annotated polylines: exact intersection, vertical-line branch, parallel verticals
    l1 = np.array([[0., 0.], [10., 10.], [20., 20.], [30., 30.]])
    l2 = np.array([[0., 30.], [10., 20.], [30., 0.]])
    assert np.allclose(an.line_intersection(l1, l2), (15.0, 15.0))
    v = np.array([[12.0, 0.], [12.2, 50.], [11.9, 100.]])
    xv, yv = an.line_intersection(v, l1)
    assert abs(xv - yv) < 1e-9 and 11.9 <= xv <= 12.2                          # on l1 (y = x), x from the near-vertical polyline
    assert an.line_intersection(v, v + np.array([5.0, 0.0])) is None
    assert an.inside((961.0, 10.0)) is None and an.inside((961.0, 10.0), margin=2.0) == (961.0, 10.0) and an.inside(None) is None
    # homography from exact correspondences with one gross outlier
    rng = np.random.default_rng(0)
    src = rng.uniform(-50, 50, (12, 2))
    H = np.array([[8.0, 1.0, 480.0], [-0.5, 7.0, 270.0], [1e-3, 2e-3, 1.0]])
    h = np.c_[src, np.ones(12)] @ H.T
    dst = h[:, :2] / h[:, 2:3]
    dst[3] += 80.0
    Hm = an.homography_ransac(src, dst, 5.0)
    assert np.abs(Hm / Hm[2, 2] - H).max() < 1e-6


def test_synthetic_annotation_recovers_the_template_projection():
    """End to end on the generator: labels from clicked lines / circles land on the projected template points."""
    from sncal_amd.pitch import PITCH_ARRAY
    hits = total = 0
    for seed in range(6):
        pts, cam = sncal_amd.synth.synthetic_annotation(seed, noise_px=0.0)
        labels, _ = an.get_intersections(pts)
        proj = cam.project_points(PITCH_ARRAY)
        for i, p in labels.items():
            if p is not None and i not in an.NOT_ON_PLANE:
                total += 1
                hits += np.hypot(p[0] - proj[i, 0], p[1] - proj[i, 1]) < 1.0
    assert total > 60 and hits / total > 0.97, (hits, total)
