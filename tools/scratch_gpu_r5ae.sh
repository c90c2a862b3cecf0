#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
for S in 8 1 4 16 64; do
  echo "== interleave $S"; SNCAL_TT_QUEUE_INTERLEAVE=$S DEV_TOP=1 timeout 600 python tools/dev_bench.py 64 fp16x3 6 2>&1 | grep -v amdgpu.ids | head -2
done
echo "== r4"; SNCAL_LIB_PATH=tools/ab/libsncal_r4.so DEV_TOP=1 timeout 600 python tools/dev_bench.py 64 fp16x3 6 2>&1 | grep -v amdgpu.ids | head -2
cd /tmp
for S in 8 1; do
rm -rf /tmp/pf; SNCAL_TT_QUEUE_INTERLEAVE=$S rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pf -- python $GRAFT_REPO_ROOT/tools/dev_bench.py 64 fp16x3 1 > /dev/null 2>&1
python - $S <<'PY'
import csv, glob, sys, collections
f = glob.glob('/tmp/pf/**/*counter_collection.csv', recursive=True)[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if r['Counter_Name'] == 'FETCH_SIZE' and 'conv_tt_kernel<2>' in r['Kernel_Name'] and 'c32' in r['Kernel_Name']: d['tt'].append(float(r['Counter_Value']))
v = d['tt']; print('interleave', sys.argv[1], 'conv_tt c32 launches', len(v), 'FETCH_SIZE KB mean', sum(v) / len(v))
PY
done
