#!/bin/bash
# round 6 parity evidence: (a) trained-like checkpoint (four decades of block-internal scales) with the C-side rebalancing and without it,
# (b) 16384 frames of the deep-path workload, fp16x3 vs the exact-fp32 engine
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6p; mkdir -p $O; cd $R
PARITY_TRAINED_LIKE=0 timeout 900 python tools/parity_large.py 512 0.35 > $O/trainedlike_512_equalized.log 2>&1; cp gpurun_out/parity_large_512.json $O/trainedlike_512_equalized.json 2>/dev/null
PARITY_TRAINED_LIKE=0 PARITY_NO_EQUALIZE=1 timeout 900 python tools/parity_large.py 512 0.35 > $O/trainedlike_512_no_equalization.log 2>&1; cp gpurun_out/parity_large_512.json $O/trainedlike_512_no_equalization.json 2>/dev/null
timeout 2400 python tools/parity_large.py 16384 0.35 > $O/parity_large_16384.log 2>&1; cp gpurun_out/parity_large_16384.json $O/ 2>/dev/null
grep -v amdgpu.ids $O/trainedlike_512_equalized.log | tail -4; grep -v amdgpu.ids $O/trainedlike_512_no_equalization.log | tail -4; grep -v amdgpu.ids $O/parity_large_16384.log | tail -6
