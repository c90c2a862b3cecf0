#!/bin/bash
# A/B: bblockx3 with 8 multiplying waves (2 per SIMD, J = 2) vs 4 (1 per SIMD, J = 4): tools/ab/libsncal_nw4.so = -DBBX_NW=4
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6c; mkdir -p $O; cd $R
hipcc --offload-arch=gfx950 -O2 tools/dev/mfma_rate.hip -o /tmp/mfma_rate && /tmp/mfma_rate > $O/mfma_rate.txt 2>&1; cat $O/mfma_rate.txt
for lib in "" tools/ab/libsncal_nw4.so; do
  tag=$( [ -z "$lib" ] && echo nw8 || echo nw4 )
  export SNCAL_LIB_PATH=$( [ -z "$lib" ] && echo "" || echo $R/$lib ); [ -z "$SNCAL_LIB_PATH" ] && unset SNCAL_LIB_PATH
  for rep in 1 2; do DEV_TOP=4 python tools/dev_bench.py 64 fp16x3 3 2>&1 | grep -v "^W" | head -5; done > $O/dev_$tag.txt
  cat $O/dev_$tag.txt
  SNCAL_BBX_TRACE=$O/bbx_$tag.bin python tools/dev/bbx_trace_run.py > /dev/null 2>&1; python tools/bbx_trace.py $O/bbx_$tag.bin 2>&1 | tail -4; rm -f $O/bbx_$tag.bin
done
export SNCAL_LIB_PATH=$R/tools/ab/libsncal_nw4.so
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "fp16x3" 2>&1 | tail -3
timeout 600 python -m pytest tests/test_hrnet_gpu.py -x -q 2>&1 | tail -3
