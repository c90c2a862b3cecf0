#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_range_guard_gpu.py -m gpu -q -s 2>&1 | grep -v "^$" | tail -25
SNCAL_BENCH_DIAG=nosolve timeout 600 python bench.py --steps 20 --warmup 5 2>&1 | grep diag
SNCAL_LIB_PATH=tools/ab/libsncal_r4.so SNCAL_BENCH_DIAG=nosolve timeout 600 python bench.py --steps 20 --warmup 5 2>&1 | grep diag
SNCAL_BENCH_DIAG=nosolve timeout 600 python bench.py --steps 20 --warmup 5 2>&1 | grep diag
SNCAL_LIB_PATH=tools/ab/libsncal_r4.so SNCAL_BENCH_DIAG=nosolve timeout 600 python bench.py --steps 20 --warmup 5 2>&1 | grep diag
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r5s_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r5s_pytest.log
