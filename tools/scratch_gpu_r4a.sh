#!/bin/bash
O=gpurun_out/r4q; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; tail -12 $O/pytest.txt
