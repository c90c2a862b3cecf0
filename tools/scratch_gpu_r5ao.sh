#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
for B in 1 4 16; do for lib in "" tools/ab/libsncal_r4.so; do
  echo "== B $B lib ${lib:-main}"; SNCAL_LIB_PATH=$lib DEV_TOP=4 timeout 300 python tools/dev_bench.py $B fp16x3 6 2>&1 | grep -v amdgpu.ids | grep "bblock\|ms/step,"
done; done
for rep in 1 2; do for lib in "" tools/ab/libsncal_r4.so; do
  echo "== B 64 lib ${lib:-main}"; SNCAL_LIB_PATH=$lib DEV_TOP=4 timeout 300 python tools/dev_bench.py 64 fp16x3 6 2>&1 | grep -v amdgpu.ids | grep "bblock\|ms/step,"
done; done
echo "== 1080p B 16"; for lib in "" tools/ab/libsncal_r4.so; do SNCAL_LIB_PATH=$lib DEV_H=1080 DEV_W=1920 DEV_TOP=4 timeout 300 python tools/dev_bench.py 16 fp16x3 3 2>&1 | grep -v amdgpu.ids | grep "bblock\|ms/step,"; done
