"""Compare per-kernel average durations of the bench step with and without the side-stream solves (tools/dev/solve_interference.sh),
and show where the solve kernels sit in time relative to the head kernel."""
import csv
import sys

O = sys.argv[1]


def stats(mode):
    out = {}
    for r in csv.DictReader(open(f'{O}/stats_{mode}.csv')):
        out[r['Name']] = (int(r['Calls']), float(r['TotalDurationNs']) / 1e6)
    return out


s = {m: stats(m) for m in ('base', 'nosolve', 'old')}
names = sorted(s['nosolve'], key=lambda n: -s['nosolve'][n][1])
print(f"{'kernel':70s} {'calls':>6s} {'nosolve ms':>11s} {'base ms':>9s} {'old ms':>9s}")
tot = {m: 0.0 for m in s}
for n in names:
    if 'voter' in n or 'calibrate' in n:
        print(f'{n[:70]:70s} ' + ' '.join(f'{m}: {s[m].get(n, (0, 0.0))[0]} calls {s[m].get(n, (0, 0.0))[1]:.1f} ms' for m in ('base', 'old')))
        continue
    row = [s[m].get(n, (0, 0.0)) for m in ('nosolve', 'base', 'old')]
    for m, r in zip(('nosolve', 'base', 'old'), row):
        tot[m] += r[1]
    if row[0][1] > 1.0:
        print(f'{n[:70]:70s} {row[0][0]:6d} {row[0][1]:11.2f} {row[1][1]:9.2f} {row[2][1]:9.2f}')
print('sum of network kernels (ms over the run):', {m: round(v, 1) for m, v in tot.items()})

for mode in ('base', 'old'):
    rows = list(csv.DictReader(open(f'{O}/trace_{mode}.csv')))
    key_s = 'Start_Timestamp' if 'Start_Timestamp' in rows[0] else 'Start'
    key_e = 'End_Timestamp' if 'End_Timestamp' in rows[0] else 'End'
    heads = [(int(r[key_s]), int(r[key_e])) for r in rows if 'headx3' in r['Kernel_Name']]
    solves = [(int(r[key_s]), int(r[key_e]), r['Kernel_Name'][22:40]) for r in rows if 'voter_' in r['Kernel_Name'] or 'calibrate_kernel' in r['Kernel_Name']]
    print(mode, 'heads', len(heads), 'solve kernels', len(solves))
    for hs, he in heads[-6:-1]:
        near = [(round((a - hs) / 1e6, 2), round((b - hs) / 1e6, 2), n) for a, b, n in solves if hs - 60e6 < a < he + 60e6]
        print(f'  head 0 .. {(he - hs) / 1e6:.2f} ms; solve kernels (start, end relative to the head start):', near)
