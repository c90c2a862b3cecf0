#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_hrnet_gpu.py -m gpu -x -q 2>&1 | tail -3
for B in 1 4 8; do for lib in "" tools/ab/libsncal_r4.so; do
  echo "== B $B lib ${lib:-main}"; SNCAL_LIB_PATH=$lib DEV_TOP=4 timeout 300 python tools/dev_bench.py $B fp16x3 6 2>&1 | grep -v amdgpu.ids | grep "bblock\|ms/step,"
done; done
