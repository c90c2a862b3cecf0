#!/bin/bash
# usage (GPU box): tools/dev/solve_interference.sh TAG -- which kernels pay for the side-stream solves: rocprofv3 kernel stats of the
# bench step with the solves (base), without them (SNCAL_BENCH_DIAG=nosolve) and with the four-wave voter of rounds 1-3 (old), same box; tools/dev/solve_interference.py compares them.
tag=${1:-si}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for mode in base nosolve old; do
  env=""
  [ $mode = nosolve ] && env="SNCAL_BENCH_DIAG=nosolve"
  [ $mode = base ] && env="SNCAL_SOLVE_TASKS=1"
  [ $mode = old ] && env="SNCAL_SOLVE_TASKS=0"
  env $env rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$mode -o trace -- python $R/bench.py --no-cpu-baseline --no-parity > $O/$mode.out 2> $O/$mode.err
  find $O/prof_$mode -name "*kernel_stats.csv" -exec cp {} $O/stats_$mode.csv \;
  find $O/prof_$mode -name "*kernel_trace.csv" -exec cp {} $O/trace_$mode.csv \;
  rm -rf $O/prof_$mode
done
cd $R
python tools/dev/solve_interference.py $O
