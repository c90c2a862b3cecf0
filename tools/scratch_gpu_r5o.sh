#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q 2>&1 | tail -3
SNCAL_BENCH_DIAG=noprof timeout 600 python bench.py --steps 20 --warmup 5 2>&1 | grep diag
SNCAL_BENCH_DIAG=nosolve timeout 600 python bench.py --steps 20 --warmup 5 2>&1 | grep diag
SNCAL_BENCH_DIAG=noprof SNCAL_BENCH_REFINE_CAP=200 timeout 600 python bench.py --steps 20 --warmup 5 2>&1 | grep diag
SNCAL_BENCH_DIAG=noprof SNCAL_SOLVE_CUS_PER_XCD=0 timeout 600 python bench.py --steps 20 --warmup 5 2>&1 | grep diag
SNCAL_BENCH_DIAG=noprof timeout 600 python bench.py --steps 40 --warmup 5 2>&1 | grep diag
timeout 300 python tools/dev/pipe_timeline.py 24 2>&1 | grep -v amdgpu.ids | head -1
SNCAL_BENCH_REFINE_CAP=200 timeout 300 python tools/dev/pipe_timeline.py 24 2>&1 | grep -v amdgpu.ids | head -1
