#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
for B in 1 8 16; do for lib in "" tools/ab/libsncal_r4.so; do
  echo "== B $B lib ${lib:-main}"; SNCAL_LIB_PATH=$lib DEV_TOP=2 timeout 300 python tools/dev_bench.py $B fp16x3 6 2>&1 | grep -v amdgpu.ids | head -3
done; done
