#!/bin/bash
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_hrnet_gpu.py -m gpu -x -q -k "fp16x3" 2>&1 | tail -2
for v in new prev new prev; do
if [ $v = new ]; then unset SNCAL_LIB_PATH; else export SNCAL_LIB_PATH=tools/ab/libsncal_prev.so; fi
echo "--- $v"; DEV_TOP=9 timeout 300 python tools/dev_bench.py 64 fp16x3 5 2>&1 | grep "conv<\|ms/step"
done
