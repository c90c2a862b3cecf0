"""Dev helper: summarise a SNCAL_CONV_TRACE dump (16 x u64 per workgroup: hwid, start, [wait_done, mfma_done] x chunks, end)."""
import sys, numpy as np
a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 16)
nch = min(int(sys.argv[2]) if len(sys.argv) > 2 else 3, 6)      # (the dump holds 16 stamps per workgroup: at most six chunks' pairs)
a = a[a[:, 1] > 0]
t0 = a[:, 1].min()
T = (a[:, 1:] - t0).astype(np.int64)          # column k = stamp k+1
start, end = T[:, 0], T[:, 14]
print('workgroups', len(a), 'kernel span (clk)', int(end.max()), 'mean WG lifetime', int((end - start).mean()))
prev = start
for c in range(nch):
    w, m = T[:, 1 + 2 * c], T[:, 2 + 2 * c]
    print(f' chunk {c}: stage wait {int((w - prev).mean()):6d} (p10 {int(np.percentile(w - prev, 10))}, p90 {int(np.percentile(w - prev, 90))})   mfma {int((m - w).mean()):6d} (p10 {int(np.percentile(m - w, 10))}, p90 {int(np.percentile(m - w, 90))})')
    prev = m
print(f' epilogue {int((end - prev).mean()):6d} (p10 {int(np.percentile(end - prev, 10))}, p90 {int(np.percentile(end - prev, 90))})')
hw = (a[:, 0] >> np.uint64(32)).astype(np.int64); xcc = (a[:, 0] & np.uint64(0xf)).astype(np.int64)
cu = ((hw >> 8) & 0xf) | (((hw >> 13) & 0x7) << 4) | (xcc << 8)
if T[:, 9].max() > 0:
    print(' epilogue: barrier %d, bias+stage writes %d, rest %d' % (int((T[:, 9] - prev).mean()), int((T[:, 10] - T[:, 9]).mean()), int((end - T[:, 10]).mean())))
if T[:, 11].max() > 0:
    print(' chunk 1 round: barrier+issue %d, vmcnt wait %d, barrier %d' % (int((T[:, 11] - T[:, 2]).mean()), int((T[:, 12] - T[:, 11]).mean()), int((T[:, 3] - T[:, 12]).mean())))
print(' distinct CUs', len(np.unique(cu)))
# overlap on one CU: fraction of a WG's mfma time during which another WG on the same CU is also in an mfma phase
tot = both = 0
for u in np.unique(cu)[:64]:
    idx = np.where(cu == u)[0]
    iv = [(T[i, 1 + 2 * c], T[i, 2 + 2 * c], i) for i in idx for c in range(nch)]
    for (s1, e1, i1) in iv:
        tot += e1 - s1
        for (s2, e2, i2) in iv:
            if i2 != i1: both += max(0, min(e1, e2) - max(s1, s2))
print(' mfma-phase time overlapped with another WG\'s mfma phase on the same CU: %.0f%%' % (100.0 * both / max(tot, 1)))
# CU timeline utilisation: union of mfma intervals / span
u0 = np.unique(cu)[0]; idx = np.where(cu == u0)[0]
ev = sorted((T[i, 1 + 2 * c], T[i, 2 + 2 * c]) for i in idx for c in range(nch))
un = 0; cs, ce = ev[0]
for s, e in ev[1:]:
    if s > ce: un += ce - cs; cs, ce = s, e
    else: ce = max(ce, e)
un += ce - cs
print(' CU %d: %d WGs, union of mfma phases %d clk of %d span (%.0f%%)' % (u0, len(idx), un, int(end[idx].max() - start[idx].min()), 100.0 * un / (end[idx].max() - start[idx].min())))
for i in idx[np.argsort(start[idx])][:8]:
    print('   WG', i, 'start', start[i], 'stamps', ' '.join(str(int(x)) for x in T[i, 1:1 + 2 * nch]), 'end', end[i])
