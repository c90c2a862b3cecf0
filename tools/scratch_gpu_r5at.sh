#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
for abl in 0 1 2 4 3 5 6; do
  echo "== ablate $abl"; SNCAL_TT_ABLATE=$abl DEV_TOP=1 timeout 300 python tools/dev_bench.py 64 fp16x3 4 2>&1 | grep -v amdgpu.ids | grep "conv_tt"
done
SNCAL_TT_TRACE=/tmp/tt64.bin timeout 120 python tools/dev/tt_trace_run.py fp16x3 64 > /dev/null 2>&1; python tools/tt_trace.py /tmp/tt64.bin 2>&1 | tail -30
