#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
timeout 300 python tools/dev/crawl_lm_probe.py 2>&1 | grep -v amdgpu.ids | grep "iterations:"
SNCAL_LIB_PATH=tools/ab/libsncal_lmt.so timeout 300 python tools/dev/crawl_lm_probe.py 2>&1 | grep "LM W" | tail -1
timeout 300 python tools/dev/lm_iter_probe.py 2>&1 | grep -v amdgpu.ids
timeout 1200 python -m pytest tests/test_solve_gpu.py -m gpu -x -q 2>&1 | tail -3
