#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
timeout 900 python -m pytest tests/test_range_guard_gpu.py -m gpu -q -s 2>&1 | grep -v "^$" | tail -30
