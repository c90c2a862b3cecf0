#!/bin/bash
for i in 1 2; do
echo "--- new"; DEV_TOP=3 timeout 300 python tools/dev_bench.py 64 fp16x3 5 2>&1 | grep -v amdgpu.ids
echo "--- prev"; SNCAL_LIB_PATH=tools/ab/libsncal_prev.so DEV_TOP=3 timeout 300 python tools/dev_bench.py 64 fp16x3 5 2>&1 | grep -v amdgpu.ids
done
