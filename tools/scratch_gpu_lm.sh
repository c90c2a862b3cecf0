#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
timeout 300 python tools/dev/crawl_lm_probe.py 2>&1 | grep -v amdgpu.ids | grep "iterations:\|hash"
timeout 300 python tools/dev/lm_iter_probe.py 2>&1 | grep -v amdgpu.ids
timeout 1200 python -m pytest tests/test_solve_gpu.py -m gpu -x -q 2>&1 | tail -3
