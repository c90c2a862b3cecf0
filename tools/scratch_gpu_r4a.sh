#!/bin/bash
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "fp16x3_engine_w48_540p" 2>&1 | tail -2
for v in new prev new prev; do
if [ $v = new ]; then unset SNCAL_LIB_PATH; else export SNCAL_LIB_PATH=tools/ab/libsncal_prev.so; fi
echo "--- $v"; DEV_TOP=1 timeout 300 python tools/dev_bench.py 64 fp16x3 5 2>&1 | grep "conv_tt\|ms/step"
done
