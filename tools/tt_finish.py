#!/usr/bin/env python3
"""Finish-time spread of the teams of one conv_tt launch (SNCAL_TT_TRACE dump: slot 254 = team start, 255 = team finish on the chip-wide
100 MHz s_memrealtime clock; the s_memtime stamps of the other slots are per-CU clocks and only compare inside a workgroup).
A persistent launch ends when its LAST team ends: 1 - mean / max is the share of team-time that perfect balancing could still save."""
import sys
import numpy as np
t = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 256).astype(np.int64)
start, fin = t[:, 254], t[:, 255]
ok = (start > 0) & (fin > 0)
t0 = start[ok].min()
f = (fin[ok] - t0) / 100.0            # microseconds
s = (start[ok] - t0) / 100.0
print(f'{ok.sum()} teams; start spread {s.max():.1f} us; finish min {f.min():.1f}  p10 {np.percentile(f, 10):.1f}  median {np.median(f):.1f}  p90 {np.percentile(f, 90):.1f}  max {f.max():.1f} us')
print(f'mean team finishes at {f.mean() / f.max():.3f} of the launch; (max - median) / max = {(f.max() - np.median(f)) / f.max():.3f}')
idx = np.flatnonzero(ok)
for k in range(8):
    sel = f[(idx // 2) % 8 == k]
    print(f'  XCD {k}: finish median {np.median(sel):.1f}  max {sel.max():.1f} us')
