#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_range_guard_gpu.py -m gpu -q -s 2>&1 | grep -v "^$" | tail -30
timeout 1500 python -m pytest tests -m gpu -q --deselect tests/test_range_guard_gpu.py > gpurun_out/r5t_pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r5t_pytest.log
