#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6b; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_range_guard_gpu.py tests/test_solve_scipy_gpu.py tests/test_dist_gpu.py tests/test_solve_gpu.py tests/test_pipeline_gpu.py -x -q 2>&1 | tail -8
cd /tmp; export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o trace -- python $R/tools/dev/c4_solve_probe.py > $O/probe.log 2>&1
cat $O/probe.log | grep -v "^W\|rocprof" | tail -20
find $O/prof -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
find $O/prof -name "*.csv" -size +1M -delete
grep -i "first_pass\|voter\|calibrate" $O/kernel_stats.csv | cut -c1-200
