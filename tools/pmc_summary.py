"""Dev helper: per-kernel mean of every counter in a rocprofv3 --pmc output directory."""
import csv, glob, sys, collections, re
d = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        m = re.search(r'conv_(?:group_)?kernelI(\w+?)Li(\d)ELi(\d)ELi(\d)ELi(\d)ELi(\d)E', k)
        if 'conv_s2p_kernel' in k:
            k, m = 'conv_s2p<fp16x3,k3,s2>', None
        elif 'conv_shared_s2_kernel' in k:
            k, m = 'conv_shared_s2<fp16x3,k3,NI2,G3>', None
        elif 'conv_tt_kernel' in k:
            mode = 'fp8' if ('ILb1' in k or '<true>' in k or 'ILi1E' in k or 'kernel<1>' in k) else 'fp16x3' if ('ILi2E' in k or 'kernel<2>' in k) else 'bf16'
            k = 'conv_tt<%s,k3,s1,%s>' % (mode, '12x32x64' if 'c23' in k else '4x32x96' if 'c31' in k else '8x32x96')
        elif m:
            ty = {'DF16b': 'bf16', 'NS_4x3_tE': 'fp16x3'}.get(m.group(1), 'f32')
            k = 'conv<%s,k%s,s%s,NI%s,MI%s,G%s>' % ((ty,) + m.groups()[1:])
        else:
            m2 = re.search(r'conv_(?:group_)?kernel<(.*?), (\d), (\d), (\d), (\d), (\d)>', k)
            if m2:
                ty = 'bf16' if 'bf16' in m2.group(1) else 'fp16x3' if 'x3_t' in m2.group(1) else 'f32'
                k = 'conv<%s,k%s,s%s,NI%s,MI%s,G%s>' % ((ty,) + m2.groups()[1:])
            else:
                k = ('headx3_fused' if 'headx3_kernel' in k else 'head_fused' if ('head_fused' in k or 'head32_kernel' in k) else
                     'bneck_tail_ds_x3' if ('bneck_pair_kernel' in k and ('ILb1ELb0E' in k or '<true, false>' in k)) else 'bneck_seam_x3' if 'bneck_pair_kernel' in k else
                     'bblockx3_fused' if 'bblockx3_kernel' in k else 'bblock48_fused' if 'bblock48_kernel' in k else 'upsample_add' if 'upsample_add_kernel' in k else k[:48])
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
