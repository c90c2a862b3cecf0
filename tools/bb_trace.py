#!/usr/bin/env python3
"""Summarise a SNCAL_BB_TRACE dump (bblock.hip): per wave of every workgroup, clocks per phase summed over its tiles."""
import sys
import numpy as np
NW = 8
t = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, NW, 8).astype(np.float64)
n = t[:, :, 5]
names = ['halo wait + opening barrier + stores issued', 'conv1 MFMAs', 'residual read, mid write', 'conv2 MFMAs', 'epilogue arithmetic', None,
         'barrier after the mid write', 'next halo requested']
order = [0, 1, 2, 6, 7, 3, 4]
tot = t[:, :, :5].sum(axis=2) + t[:, :, 6] + t[:, :, 7]
print(f'{t.shape[0]} workgroups x {NW} waves, tiles per workgroup {n.min():.0f}..{n.max():.0f}; clocks per tile, mean over workgroups, per wave:')
print(' ' * 46 + ''.join(f'  wave{w}' for w in range(NW)))
for k in order:
    print(f'  {names[k]:44s}' + ''.join(f'{np.mean(t[:, w, k] / n[:, w]):7.0f}' for w in range(NW)))
print(f'  {"total":44s}' + ''.join(f'{np.mean(tot[:, w] / n[:, w]):7.0f}' for w in range(NW)))
print(f'per workgroup total (wave 0) min/median/max {tot[:, 0].min():.0f} {np.median(tot[:, 0]):.0f} {tot[:, 0].max():.0f}')
