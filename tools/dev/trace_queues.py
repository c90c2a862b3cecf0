"""Summary of a rocprofv3 --kernel-trace csv: per queue span / busy time / kernel count, the longest kernels, and for the queue that
carries the convolutions the kernels that ran longer than `factor` x their median.  python tools/dev/trace_queues.py <kernel_trace.csv>"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r['s'] = int(r['Start_Timestamp']); r['e'] = int(r['End_Timestamp'])
t0 = min(r['s'] for r in rows)
byq = collections.defaultdict(list)
for r in rows:
    byq[(r['Queue_Id'], r['Stream_Id'])].append(r)
for q, rs in sorted(byq.items()):
    rs.sort(key=lambda r: r['s'])
    busy = sum(r['e'] - r['s'] for r in rs)
    names = collections.Counter(r['Kernel_Name'].split('(')[0][-40:] for r in rs)
    print(f'queue {q}: {len(rs)} kernels, span {(rs[0]["s"] - t0) / 1e6:.1f} .. {(rs[-1]["e"] - t0) / 1e6:.1f} ms, busy {busy / 1e6:.1f} ms; top: {names.most_common(3)}')
main = max(byq.values(), key=len)
med = collections.defaultdict(list)
for r in main:
    med[r['Kernel_Name']].append(r['e'] - r['s'])
tot = collections.Counter()
for k, v in med.items():
    tot[k] = sum(v)
print('main queue, by kernel: total ms, n, median us, max us')
for k, t in tot.most_common(14):
    v = sorted(med[k])
    print(f'  {t / 1e6:9.2f} {len(v):5d} {v[len(v) // 2] / 1e3:9.1f} {v[-1] / 1e3:9.1f}  {k[:90]}')
gaps = [(main[i + 1]['s'] - main[i]['e'], main[i]['Kernel_Name'][:50], main[i + 1]['Kernel_Name'][:50]) for i in range(len(main) - 1)]
gaps.sort(reverse=True)
print('largest gaps on the main queue (ms):', [(round(g / 1e6, 2), a[-30:], b[-30:]) for g, a, b in gaps[:6]])
print('sum of gaps', sum(g for g, _, _ in gaps if g > 0) / 1e6, 'ms')
