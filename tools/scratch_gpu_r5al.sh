#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
for runs in 8 1; do
  echo "== runs $runs"; SNCAL_BBX_RUNS=$runs SNCAL_BBX_TRACE=/tmp/bbx_$runs.bin timeout 200 python tools/dev/bbx_trace_run.py > /dev/null 2>&1; python tools/bbx_trace.py /tmp/bbx_$runs.bin
done
echo "== r4 lib"; SNCAL_LIB_PATH=tools/ab/libsncal_r4.so SNCAL_BBX_TRACE=/tmp/bbx_r4.bin timeout 200 python tools/dev/bbx_trace_run.py > /dev/null 2>&1; python tools/bbx_trace.py /tmp/bbx_r4.bin
