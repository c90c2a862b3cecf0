import sys, numpy as np
a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 16); a = a[a[:, 1] > 0]
T = (a[:, 1:12].astype(np.int64) - a[:, 1:2].astype(np.int64))
names = ['round0 (x + W1c0 + W1c1)', 'mfma conv1 c0', 'barrier + issue W2c0', 'mfma conv1 c1', 'residual->regs, mid write', 'round (W2c0, mid) + issue W2c1', 'mfma conv2 c0', 'round W2c1', 'mfma conv2 c1', 'epilogue']
print('workgroups', len(a), 'mean lifetime', int(T[:, 10].mean()))
for k, nm in enumerate(names):
    d = T[:, k + 1] - T[:, k]
    print(f' {nm:28s} {int(d.mean()):6d} (p10 {int(np.percentile(d, 10))}, p90 {int(np.percentile(d, 90))})')
