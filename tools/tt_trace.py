#!/usr/bin/env python3
"""Summarise a SNCAL_TT_TRACE dump (conv_tt.hip: six s_memtime stamps per stage and team):
[0] LOAD begins  [1] epilogue/setup done  [2] DMA issued  [3] DMA landed  [4] MULTIPLY begins  [5] MFMAs issued."""
import sys
import numpy as np

t = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 256).astype(np.int64)
ep = t[:, 192:]
t = t[:, :192]
rows = []
erows = []
for team in range(ep.shape[0]):
    v = ep[team][ep[team] > 0]
    n = len(v) // 6
    if n:
        erows.append(np.diff(v[:n * 6].reshape(n, 6), axis=1))
if erows:
    e = np.concatenate(erows)
    for i, nm in enumerate(['epi: residual loads issued', 'epi: tile row 0 (incl. wait for the residual)', 'epi: tile row 1',
                            'setup: halo offsets', 'setup: accumulators from the LDS bias table']):
        print(f'  {nm:48s} median {np.median(e[:, i]):8.0f}  mean {e[:, i].mean():8.0f}  p90 {np.percentile(e[:, i], 90):8.0f}')
for team in range(t.shape[0]):
    v = t[team][t[team] > 0]
    n = len(v) // 6
    if n < 2:
        continue
    v = v[:n * 6].reshape(n, 6)
    d = np.diff(v, axis=1)                       # epi/setup, issue, wait, barrier, mfma
    nxt = v[1:, 0] - v[:-1, 5]                    # closing barrier of the multiply phase
    rows.append(np.c_[d[:-1], nxt, v[1:, 0] - v[:-1, 0]])
a = np.concatenate(rows)
names = ['epi+setup', 'dma issue', 'dma wait', 'barrier(load)', 'mfma', 'barrier(mult)', 'stage period']
print(f'{len(rows)} teams, {len(a)} stages; clocks per stage (s_memtime ticks = shader clocks)')
for i, nm in enumerate(names):
    c = a[:, i]
    print(f'  {nm:14s} median {np.median(c):8.0f}  mean {c.mean():8.0f}  p90 {np.percentile(c, 90):8.0f}')
first = t[:, 0][t[:, 0] > 0]
last = t.max(axis=1)[t.max(axis=1) > 0]
print('kernel span (first stamp -> last stamp):', int(last.max() - first.min()), 'clocks; team finish spread p10/p50/p90:',
      [int(np.percentile(last - first.min(), q)) for q in (10, 50, 90)])
