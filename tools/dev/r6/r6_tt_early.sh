#!/bin/bash
# A/B on one box: conv_tt's early halo request in front of fp32-storing epilogues (default) vs SNCAL_TT_ABLATE=128 (off)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6d; mkdir -p $O; cd $R
for rep in 1 2; do
  for ab in 0 128; do
    SNCAL_TT_ABLATE=$ab DEV_TOP=2 python tools/dev_bench.py 64 fp16x3 4 2>&1 | grep -v "^W\|amdgpu.ids" | head -3 | sed "s/^/ablate=$ab /"
  done
done | tee $O/ab.txt
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "fp16x3" 2>&1 | tail -3
timeout 600 python -m pytest tests/test_hrnet_gpu.py tests/test_configs_gpu.py -x -q 2>&1 | tail -3
