#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
timeout 200 python tools/dev_bench.py 2 fp16x3 2 2>&1 | grep -v amdgpu.ids | head -1 || echo "SMALL RUN FAILED/HUNG"
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_hrnet_gpu.py -m gpu -x -q 2>&1 | tail -3
for runs in 8 1; do
  echo "== runs $runs"; SNCAL_BBX_RUNS=$runs DEV_TOP=3 timeout 300 python tools/dev_bench.py 64 fp16x3 6 2>&1 | grep -v amdgpu.ids | grep "bblock\|ms/step,"
done
SNCAL_BBX_TRACE=/tmp/bbx_8.bin timeout 200 python tools/dev/bbx_trace_run.py > /dev/null 2>&1; python tools/bbx_trace.py /tmp/bbx_8.bin
