#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-parity > /tmp/b.json 2>/tmp/b.err; echo "plain bench rc=$?"
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p2 -o t -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-parity > /tmp/t2.log 2>&1; echo "rocprofv3 + bench, masked solve streams rc=$?"; grep -c SIGSEGV /tmp/t2.log
cd $R; timeout 200 python -m pytest tests/test_pipeline_gpu.py tests/test_dist_gpu.py -x -q 2>&1 | tail -3
