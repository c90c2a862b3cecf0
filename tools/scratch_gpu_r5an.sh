#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
for dbg in 0 2 4 6 7; do
  echo "== dbg $dbg"; SNCAL_BBX_DBG=$dbg SNCAL_BBX_TRACE=/tmp/bbx_d.bin timeout 200 python tools/dev/bbx_trace_run.py > /dev/null 2>&1; python tools/bbx_trace.py /tmp/bbx_d.bin | tail -1
done
