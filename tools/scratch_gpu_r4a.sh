#!/bin/bash
O=gpurun_out/r4f; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "bf16x3_engine_w48" > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt
SNCAL_BBX_TRACE=$O/bbx.bin python tools/dev/bbx_trace_run.py > $O/log.txt 2>&1
python tools/bbx_trace.py $O/bbx.bin | tee $O/bbx_trace.txt
rm -f $O/bbx.bin
DEV_TOP=3 timeout 300 python tools/dev_bench.py 64 bf16x3 5 2>&1 | grep -v amdgpu.ids
