#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
timeout 300 python tools/dev/pipe_timeline.py 24 2>&1 | grep -v amdgpu.ids
SNCAL_BENCH_REFINE_CAP=200 timeout 300 python tools/dev/pipe_timeline.py 24 2>&1 | grep -v amdgpu.ids
SNCAL_SOLVE_CUS_PER_XCD=0 timeout 300 python tools/dev/pipe_timeline.py 24 2>&1 | grep -v amdgpu.ids
