#!/usr/bin/env python3
"""Per-CU view of a SNCAL_TT_TRACE dump: how much of a workgroup's life has a team in its MULTIPLY phase (stamps [4] -> [5]),
and what the two teams are doing when nobody multiplies.  Stamps of the two teams of a workgroup share one clock."""
import sys
import numpy as np
t = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 256).astype(np.int64)[:, :192]
NS = 6
names = ['epi+setup', 'dma issue', 'dma wait', 'arrive+token', 'mult', 'done barrier']
tot_busy = tot_span = tot_both = 0
idle_by = np.zeros((6, 6))
for b in range(t.shape[0] // 2):
    ev = []
    for k in (0, 1):
        v = t[2 * b + k]; v = v[v > 0]; n = len(v) // NS
        ev.append(v[:n * NS].reshape(n, NS))
    if len(ev[0]) < 2 or len(ev[1]) < 2:
        continue
    t0 = max(ev[0][0, 0], ev[1][0, 0]); t1 = min(ev[0][-1, 5], ev[1][-1, 5])        # both teams alive
    if t1 <= t0:
        continue
    # phase of a team at time x: index of the last stamp <= x, modulo 6
    flat = [e.reshape(-1) for e in ev]
    grid = np.arange(t0, t1, 16)
    ph = [(np.searchsorted(f, grid, side='right') - 1) % NS for f in flat]
    mult = [(p == 4) for p in ph]
    busy = mult[0] | mult[1]
    tot_busy += busy.sum(); tot_span += len(grid); tot_both += (mult[0] & mult[1]).sum()
    idle = ~busy
    np.add.at(idle_by, (ph[0][idle], ph[1][idle]), 1)
print(f'a team multiplying: {tot_busy / tot_span:.3f} of the time both teams are alive (both at once: {tot_both / tot_span:.3f})')
idle_by /= idle_by.sum()
print('when nobody multiplies, (team A phase, team B phase) shares:')
order = np.dstack(np.unravel_index(np.argsort(-idle_by, axis=None), idle_by.shape))[0]
for i, j in order[:10]:
    print(f'  {names[i]:14s} | {names[j]:14s}  {idle_by[i, j]:.3f}')

# ---- launch-level view: every workgroup starts with the launch, so the kernel lasts as long as the longest workgroup (the clocks
# of different CUs are not comparable, spans are)
tt = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 256).astype(np.int64)[:, :192]
spans, mults, stages = [], [], []
for b in range(tt.shape[0] // 2):
    lo, hi, m, ns = None, None, 0, 0
    for k in (0, 1):
        v = tt[2 * b + k]; v = v[v > 0]; n = len(v) // NS
        if n == 0:
            continue
        v = v[:n * NS].reshape(n, NS)
        lo = v[0, 0] if lo is None else min(lo, v[0, 0]); hi = v[-1, 5] if hi is None else max(hi, v[-1, 5])
        m += (v[:, 5] - v[:, 4]).sum(); ns += n
    if lo is not None:
        spans.append(hi - lo); mults.append(m); stages.append(ns)
spans, mults, stages = np.array(spans), np.array(mults), np.array(stages)
print(f'workgroup span (first stamp -> last MFMA) min/p10/p50/p90/max: {spans.min()} {int(np.percentile(spans, 10))} {int(np.median(spans))} {int(np.percentile(spans, 90))} {spans.max()}')
print(f'stages per workgroup min/median/max: {stages.min()} {int(np.median(stages))} {stages.max()} (a trace holds 32 stages per team)')
print(f'multiply time / longest span: {mults.mean() / spans.max():.3f}   multiply time / own span (mean): {(mults / spans).mean():.3f}')
for x in range(8):
    print(f'  XCD {x}: span max {spans[x::8].max()}  median {int(np.median(spans[x::8]))}  stages median {int(np.median(stages[x::8]))}')
