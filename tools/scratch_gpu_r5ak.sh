#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
timeout 200 python tools/dev_bench.py 2 fp16x3 2 2>&1 | grep -v amdgpu.ids | head -2 || echo "SMALL RUN FAILED/HUNG"
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_hrnet_gpu.py -m gpu -x -q 2>&1 | tail -4
for runs in 8 1 4; do
  echo "== runs $runs"; SNCAL_BBX_RUNS=$runs DEV_TOP=3 timeout 300 python tools/dev_bench.py 64 fp16x3 6 2>&1 | grep -v amdgpu.ids | head -4
done
echo "== r4 lib"; SNCAL_LIB_PATH=tools/ab/libsncal_r4.so DEV_TOP=3 timeout 300 python tools/dev_bench.py 64 fp16x3 6 2>&1 | grep -v amdgpu.ids | head -4
for B in 1 8; do echo "== B $B"; DEV_TOP=3 timeout 300 python tools/dev_bench.py $B fp16x3 6 2>&1 | grep -v amdgpu.ids | grep "bblock\|ms/step,"; done
