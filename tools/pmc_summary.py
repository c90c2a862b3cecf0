"""Dev helper: per-kernel mean of every counter in a rocprofv3 --pmc output directory."""
import csv, glob, sys, collections, re
d = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        m = re.search(r'conv_(?:group_)?kernelI(\w+?)Li(\d)ELi(\d)ELi(\d)ELi(\d)ELi(\d)E', k)
        if 'conv_tt_kernel' in k: k = 'conv_tt<%s,k3,s1,8x32x96>' % ('fp8' if ('ILb1' in k or '<true>' in k) else 'bf16')
        elif m: k = 'conv<%s,k%s,s%s,NI%s,MI%s,G%s>' % ((('bf16' if m.group(1) == 'DF16b' else 'f32'),) + m.groups()[1:])
        else:
            m2 = re.search(r'conv_(?:group_)?kernel<.*?(\d), (\d), (\d), (\d)>', k)
            k = 'conv<NI%s,MI%s,G%s>' % m2.groups()[:3] if m2 else ('head_fused' if ('head_fused' in k or 'head32_kernel' in k) else 'bblock48_fused' if 'bblock48_kernel' in k else k[:48])
        acc[k][r['Counter_Name']] += float(r['Counter_Value'])
        key = (r['Dispatch_Id'], k)
        if key not in seen: seen.add(key); cnt[k] += 1
lines = []
for k, c in sorted(acc.items(), key=lambda kv: -sum(kv[1].values()))[:8]:
    lines.append('%-34s n=%4d ' % (k, cnt[k]) + ' '.join('%s=%.4g' % (n, v / cnt[k]) for n, v in sorted(c.items())))
import json
json.dump({k: {n: v / cnt[k] for n, v in c.items()} for k, c in acc.items()}, open(d + '/summary.json', 'w'), indent=1)
open(d + '/summary.txt', 'w').write('\n'.join(lines) + '\n')
print('\n'.join(lines))
