#!/bin/bash
O=gpurun_out/r4d; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "bf16x3_engine_w48_540p" > $O/pytest.txt 2>&1; tail -30 $O/pytest.txt
timeout 300 python -m pytest tests/test_hrnet_gpu.py -m gpu -x -q -k "x3" > $O/pytest2.txt 2>&1; tail -5 $O/pytest2.txt
for i in 1 2; do
echo "--- new"; DEV_TOP=8 timeout 300 python tools/dev_bench.py 64 bf16x3 5
echo "--- unfused"; SNCAL_FUSE_BBX3=0 DEV_TOP=8 timeout 300 python tools/dev_bench.py 64 bf16x3 5
done 2>&1 | grep -v amdgpu.ids | tee $O/ab.txt
