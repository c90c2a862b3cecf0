#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
timeout 120 python tools/dev_bench.py 8 fp16x3 2 2>&1 | grep -v amdgpu.ids | head -3 || echo "SMALL RUN FAILED/HUNG"
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_hrnet_gpu.py -m gpu -x -q 2>&1 | tail -3
for rep in 1 2; do for lib in "" tools/ab/libsncal_r4.so; do
  echo "== lib ${lib:-main}"; SNCAL_LIB_PATH=$lib DEV_TOP=1 timeout 300 python tools/dev_bench.py 64 fp16x3 6 2>&1 | grep -v amdgpu.ids | head -2
done; done
cd /tmp; rm -rf /tmp/pf; timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pf -- python $GRAFT_REPO_ROOT/tools/dev_bench.py 64 fp16x3 1 > /dev/null 2>&1
python - <<'PY'
import csv, glob
f = glob.glob('/tmp/pf/**/*counter_collection.csv', recursive=True)[0]
v = [float(r['Counter_Value']) for r in csv.DictReader(open(f)) if r['Counter_Name'] == 'FETCH_SIZE' and 'c32' in r['Kernel_Name'] and 'conv_tt_kernel<2>' in r['Kernel_Name']]
print('pair tickets: conv_tt c32 launches', len(v), 'FETCH_SIZE KB mean', sum(v) / len(v))
PY
