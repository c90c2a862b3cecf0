#!/bin/bash
for v in 1 0 1 0; do
echo "--- mix $v"; SNCAL_TT_MIX=$v DEV_TOP=1 timeout 300 python tools/dev_bench.py 64 fp16x3 5 2>&1 | grep "conv_tt\|ms/step"
done
