#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
for st in 1 0; do echo "== stage $st"; SNCAL_HEAD_STAGE=$st SNCAL_HEAD_TRACE=/tmp/head.bin timeout 200 python tools/dev/head_trace_run.py > /dev/null 2>&1; python tools/head_trace.py /tmp/head.bin 2>&1 | tail -9; done
