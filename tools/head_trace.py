#!/usr/bin/env python3
"""Summarise a SNCAL_HEAD_TRACE dump (head32.hip): clocks of wave 0 of sampled workgroups, summed over the 25 slices."""
import sys
import numpy as np
t = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 8).astype(np.float64)
t = t[t[:, :6].sum(axis=1) > 0]
names = ['wait for the slice + barrier', 'next slice requested', 'stage 1 (13 MFMAs)', 'gather (16 LDS reads, 2 MFMAs)', 'ReLU + stage 2 (4 MFMAs)', 'prologue (boxes, B fragments)']
tot = t[:, :6].sum(axis=1)
print(f'{len(t)} sampled workgroups; clocks per workgroup (wave 0):')
for k in (5, 0, 1, 2, 3, 4):
    print(f'  {names[k]:34s} {t[:, k].mean():9.0f}   {t[:, k].mean() / tot.mean() * 100:5.1f} %')
print(f'  total {tot.mean():9.0f}   (min {tot.min():.0f}, max {tot.max():.0f})')
if t[:, 6].sum() > 0:
    print(f'  of the prologue: set-up + first requests {t[:, 6].mean():.0f}, direct fragments + wait for the staged boxes {t[:, 7].mean():.0f}, blends + splits {(t[:, 5] - t[:, 6] - t[:, 7]).mean():.0f}')
