#!/usr/bin/env python3
"""Build-generated golden cameras: the oracle's own results on seeded synthetic keypoints,
committed as tests/golden/solve_cameras.npz so that (a) the oracle cannot drift silently and (b) the GPU test
compares the HIP solve with committed numbers as well as with the live oracle.  NOT a reference-derived
fixture: OpenCV parity stays unpinned."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import solve, synth  # noqa: E402

seeds = list(range(64))
oc = solve.CameraCreatorOracle()
rec = dict(seeds=np.array(seeds), kpts=[], status=[], rmse=[], f=[], pos=[], rot=[], tag=[])
for s in seeds:
    kp, _ = synth.synth_keypoints(s, sigma_px=1.0)
    c = oc(kp, None)
    rec['kpts'].append(kp)
    rec['status'].append(0 if c is None else 1)
    rec['rmse'].append(0.0 if c is None else c.rmse)
    rec['f'].append(0.0 if c is None else c.xfocal_length)
    rec['pos'].append(np.zeros(3) if c is None else c.position)
    rec['rot'].append(np.eye(3) if c is None else c.rotation)
    rec['tag'].append('' if c is None else c.tag)
np.savez_compressed(os.path.join(ROOT, 'tests', 'golden', 'solve_cameras.npz'), **{k: np.array(v) for k, v in rec.items()})
print('solved', int(np.sum(rec['status'])), 'of', len(seeds))
