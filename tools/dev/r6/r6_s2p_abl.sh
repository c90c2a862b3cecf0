#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6h; mkdir -p $O; cd $R
DEV_TOP=8 timeout 60 python tools/dev_bench.py 2 fp16x3 1 2>&1 | grep "conv_s2p\|frames/s"; rc=${PIPESTATUS[0]}; echo "small rc $rc"; [ "$rc" != "0" ] && exit 1
for ab in ${ABL:-0 1 6 15}; do
  SNCAL_S2P_ABLATE=$ab DEV_TOP=12 timeout 90 python tools/dev_bench.py 64 fp16x3 2 2>&1 | grep "conv_s2p" | sed "s/^/ablate=$ab /"
done | tee -a $O/abl.txt
timeout 300 python -m pytest tests/test_kernels_gpu.py -x -q -k "fp16x3_engine_w48_540p" 2>&1 | tail -2
