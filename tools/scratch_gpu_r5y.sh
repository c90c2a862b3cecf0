#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
DEV_TOP=16 timeout 600 python tools/dev_bench.py 64 fp16x3 4 2>&1 | grep -v amdgpu.ids
echo "---- r4 lib"
SNCAL_LIB_PATH=tools/ab/libsncal_r4.so DEV_TOP=16 timeout 600 python tools/dev_bench.py 64 fp16x3 4 2>&1 | grep -v amdgpu.ids
