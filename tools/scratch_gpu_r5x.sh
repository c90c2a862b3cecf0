#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_range_guard_gpu.py -m gpu -q 2>&1 | tail -3
PARITY_TRAINED_LIKE=0 timeout 1200 python tools/parity_large.py 512 0.35 2>&1 | grep -v amdgpu.ids | grep "^fp16x3\|REFUSED" | cut -c1-700
PARITY_TRAINED_LIKE=1 timeout 1200 python tools/parity_large.py 512 0.1 2>&1 | grep -v amdgpu.ids | grep "^fp16x3\|REFUSED" | cut -c1-700
