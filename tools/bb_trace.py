#!/usr/bin/env python3
"""Summarise a SNCAL_BB_TRACE dump (bblock.hip): per workgroup, clocks per phase summed over its tiles."""
import sys
import numpy as np
t = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 8).astype(np.float64)
n = t[:, 5]
names = ['halo wait + opening barrier + stores issued', 'conv1 MFMAs', 'residual read, mid write, barrier, next halo requested', 'conv2 MFMAs', 'epilogue arithmetic']
tot = t[:, :5].sum(axis=1)
print(f'{len(t)} workgroups, tiles per workgroup {n.min():.0f}..{n.max():.0f}; clocks per tile (mean over workgroups):')
for k, nm in enumerate(names):
    print(f'  {nm:58s} {np.mean(t[:, k] / n):8.0f}   {np.mean(t[:, k] / tot) * 100:5.1f} %')
print(f'  total per tile {np.mean(tot / n):8.0f}; per workgroup total min/median/max {tot.min():.0f} {np.median(tot):.0f} {tot.max():.0f}')
