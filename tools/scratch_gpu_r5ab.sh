#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
timeout 1200 python -m pytest tests/test_kernels_gpu.py tests/test_hrnet_gpu.py -m gpu -x -q 2>&1 | tail -4
DEV_TOP=9 timeout 600 python tools/dev_bench.py 64 fp16x3 4 2>&1 | grep -v amdgpu.ids
echo "---- SNCAL_SHARE_S2=0"
SNCAL_SHARE_S2=0 DEV_TOP=9 timeout 600 python tools/dev_bench.py 64 fp16x3 4 2>&1 | grep -v amdgpu.ids
