#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
timeout 200 python tools/dev_bench.py 2 fp16x3 2 2>&1 | grep -v amdgpu.ids | head -1 || echo "SMALL RUN FAILED/HUNG"
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_hrnet_gpu.py tests/test_decode_gpu.py -m gpu -x -q 2>&1 | tail -3
for rep in 1 2; do for st in 1 0; do
  echo "== stage $st"; SNCAL_HEAD_STAGE=$st DEV_TOP=3 timeout 300 python tools/dev_bench.py 64 fp16x3 6 2>&1 | grep -v amdgpu.ids | grep "headx3\|ms/step,"
done; done
SNCAL_HEAD_TRACE=/tmp/head.bin timeout 200 python tools/dev/head_trace_run.py > /dev/null 2>&1; python tools/head_trace.py /tmp/head.bin 2>&1 | tail -12
