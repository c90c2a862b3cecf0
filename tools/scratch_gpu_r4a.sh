#!/bin/bash
O=gpurun_out/r4k; mkdir -p $O
timeout 1500 python -m pytest tests/test_dist_gpu.py tests/test_configs_gpu.py -m gpu -x -q -k "forced_rccl or per_gpu_share" > $O/pytest.txt 2>&1; tail -25 $O/pytest.txt
