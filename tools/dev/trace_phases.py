"""rocprofv3 kernel trace of tools/noisy_pipeline.py -> per phase (no solve / bench step / noisy step / no solve) the main queue's
per-kernel totals, so that what the concurrent solves cost each kernel can be read off one run.  python tools/dev/trace_phases.py <csv>"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r['s'] = int(r['Start_Timestamp']); r['e'] = int(r['End_Timestamp'])
byq = collections.defaultdict(list)
for r in rows:
    byq[r['Queue_Id']].append(r)
main = max(byq.values(), key=len)
main.sort(key=lambda r: r['s'])
# phases: cut at the first forward kernel after an idle gap > 60 ms or a synchronous solve
fw = [r for r in main if 'calibrate' not in r['Kernel_Name'] and 'voter' not in r['Kernel_Name'] and 'rocclr' not in r['Kernel_Name'] and 'elementwise' not in r['Kernel_Name']]
phases, cur = [], [fw[0]]
for a, b in zip(fw, fw[1:]):
    if b['s'] - a['e'] > 60e6:
        phases.append(cur); cur = []
    cur.append(b)
phases.append(cur)
def short(k):
    k = k.replace('sncal::', '').replace('(anonymous namespace)::', '').replace('void ', '')
    return k.split('(')[0][:44]
print(len(phases), 'phases of', [len(p) for p in phases], 'kernels')
tabs = []
for p in phases:
    t = collections.defaultdict(list)
    for r in p:
        t[short(r['Kernel_Name'])].append((r['e'] - r['s']) / 1e3)
    tabs.append(t)
big = [p for p in range(len(phases)) if len(phases[p]) > 500]
names = sorted(tabs[big[0]], key=lambda k: -sum(tabs[big[0]][k]))[:14]
print('kernel'.ljust(46) + ''.join(f'phase{p:>2}: total ms/ n / p50 / p90 / max us   ' for p in big))
for k in names:
    line = k.ljust(46)
    for p in big:
        v = sorted(tabs[p].get(k, [0.0]))
        line += f'{sum(v) / 1e3:8.1f} {len(v):5d} {v[len(v) // 2]:7.0f} {v[int(len(v) * 0.9)]:7.0f} {v[-1]:7.0f}   '
    print(line)
for p in big:
    span = (phases[p][-1]['e'] - phases[p][0]['s']) / 1e6
    busy = sum(r['e'] - r['s'] for r in phases[p]) / 1e6
    print(f'phase {p}: span {span:.1f} ms, busy {busy:.1f} ms, idle {span - busy:.1f} ms')
