#!/bin/bash
timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_hrnet_gpu.py -m gpu -x -q 2>&1 | tail -3
for v in new prev new prev; do
if [ $v = new ]; then unset SNCAL_LIB_PATH; else export SNCAL_LIB_PATH=tools/ab/libsncal_$v.so; fi
echo "--- $v"; DEV_TOP=5 timeout 300 python tools/dev_bench.py 64 fp16x3 5 2>&1 | grep "head\|ms/step"
done
