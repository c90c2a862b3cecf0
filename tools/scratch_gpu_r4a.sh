#!/bin/bash
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_hrnet_gpu.py -m gpu -x -q -k "fp16x3 or bf16_engine_w48_540p or independent" 2>&1 | tail -3
for v in new prev new prev; do
if [ $v = new ]; then unset SNCAL_LIB_PATH; else export SNCAL_LIB_PATH=tools/ab/libsncal_prev.so; fi
echo "--- $v"; DEV_TOP=2 timeout 300 python tools/dev_bench.py 64 fp16x3 5 2>&1 | grep "conv_tt\|ms/step"
done
for v in new prev; do
if [ $v = new ]; then unset SNCAL_LIB_PATH; else export SNCAL_LIB_PATH=tools/ab/libsncal_prev.so; fi
echo "--- bench $v"; python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'])"
done
