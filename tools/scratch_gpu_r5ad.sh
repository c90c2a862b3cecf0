#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "fp16x3" 2>&1 | tail -2
for rep in 1 2; do
for lib in "" tools/ab/libsncal_r4.so; do
  echo "== lib ${lib:-main}"; SNCAL_LIB_PATH=$lib DEV_TOP=1 timeout 600 python tools/dev_bench.py 64 fp16x3 6 2>&1 | grep -v amdgpu.ids | head -2
done; done
