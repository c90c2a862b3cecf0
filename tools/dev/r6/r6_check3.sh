#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6f; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_solve_scipy_gpu.py tests/test_line_workload_gpu.py tests/test_pipeline_gpu.py -q 2>&1 | tail -15
DEV_TOP=30 python tools/dev_bench.py 64 fp16x3 3 2>&1 | grep -v "^W\|amdgpu.ids" > $O/dev_top30.txt; cat $O/dev_top30.txt
