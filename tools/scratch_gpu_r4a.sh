#!/bin/bash
python tools/dev/golden_err.py 2>&1 | grep -v amdgpu.ids
SNCAL_LIB_PATH=tools/ab/libsncal_bf16x3.so python tools/dev/golden_err.py 2>&1 | grep "bf16x3"
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "fp16x3" 2>&1 | tail -4
