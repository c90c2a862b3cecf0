#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06v33; mkdir -p $O; cd $R
python bench.py --steps 20 --warmup 5 > $O/bench_c3.json 2> $O/bench_c3.err; echo "bench rc=$?"
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o trace -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity > $O/bench_c3_traced.json 2> $O/bench_c3_traced.err; echo "traced rc=$?"
find $O/prof -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
find $O/prof -name "*.csv" -size +2M -delete
grep -c SIGSEGV $O/bench_c3_traced.err
head -4 $O/kernel_stats.csv | cut -c1-160
