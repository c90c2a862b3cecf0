#!/bin/bash
timeout 2400 python -m pytest tests/test_kernels_gpu.py tests/test_hrnet_gpu.py tests/test_parity_gpu.py -m gpu -x -q 2>&1 | tail -8
