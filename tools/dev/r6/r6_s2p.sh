#!/bin/bash
# the pipelined stride-2 kernel: first a small guarded run (a hang must not outlive its timeout), then parity, then the A/B
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6g; mkdir -p $O; cd $R
DEV_TOP=8 timeout 120 python tools/dev_bench.py 2 fp16x3 1 2>&1 | grep -v "^W\|amdgpu.ids" | head -10; echo "small run rc ${PIPESTATUS[0]}"
[ "${PIPESTATUS[0]}" = "124" ] && exit 1
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "fp16x3" 2>&1 | tail -5
timeout 600 python -m pytest tests/test_hrnet_gpu.py -x -q 2>&1 | tail -3
for rep in 1 2; do for on in 1 0; do
  SNCAL_S2P=$on DEV_TOP=6 timeout 300 python tools/dev_bench.py 64 fp16x3 3 2>&1 | grep -v "^W\|amdgpu.ids" | head -7 | sed "s/^/s2p=$on /"
done; done | tee $O/ab.txt
