#!/usr/bin/env python3
"""Summarise a SNCAL_BBX_TRACE dump (bblockx3.hip: per multiplying wave the clocks spent in
[0] wait for the tile's x halo  [1] conv1  [2] residual read + mid write  [3] wait for everyone's mid rows  [4] conv2  [5] epilogue + stores; [7] tiles)."""
import sys
import numpy as np
t = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 8, 8).astype(np.float64)      # (workgroup, wave, slot)
t = t[t[:, 0, 7] > 0]
tiles = t[:, :, 7:8]
per = t[:, :, :6] / tiles
names = ['halo wait', 'conv1', 'res + mid write', 'mid barrier', 'conv2', 'epilogue']
print(f'{t.shape[0]} workgroups, tiles per workgroup {tiles.min():.0f}..{tiles.max():.0f}; clocks per tile and wave (median over workgroups)')
print('wave ' + ' '.join(f'{n:>16s}' for n in names) + '            total')
for w in range(8):
    m = np.median(per[:, w, :], axis=0)
    print(f'{w:4d} ' + ' '.join(f'{v:16.0f}' for v in m) + f' {m.sum():16.0f}')
m = np.median(per.reshape(-1, 6), axis=0)
print(' all ' + ' '.join(f'{v:16.0f}' for v in m) + f' {m.sum():16.0f}')
