#!/bin/bash
# usage: tools/pmc_pass.sh OUTDIR "CTR1 CTR2" "CTR3" ... -- one rocprofv3 counter pass per argument (dev helper)
out=$1; shift
cd /tmp; export TMPDIR=/tmp
for grp in "$@"; do
  tag=$(echo $grp | tr ' ' '_')
  mkdir -p $GRAFT_REPO_ROOT/gpurun_out/$out/$tag
  rocprofv3 --pmc $grp --kernel-trace -M --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$out/$tag -- python $GRAFT_REPO_ROOT/tools/dev_bench.py ${PMC_B:-32} ${PMC_DTYPE:-fp16x3} 1 > $GRAFT_REPO_ROOT/gpurun_out/$out/$tag/log.txt 2>&1
  python $GRAFT_REPO_ROOT/tools/pmc_summary.py $GRAFT_REPO_ROOT/gpurun_out/$out/$tag
  find $GRAFT_REPO_ROOT/gpurun_out/$out/$tag -name "*.csv" -size +2M -delete
done
