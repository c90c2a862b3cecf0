#!/bin/bash
timeout 900 python -m pytest tests/test_solve_gpu.py -m gpu -x -q -s -k iac 2>&1 | grep -v "^$" | tail -12
