#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
DEV_TOP=1 timeout 300 python tools/dev_bench.py 1 fp16x3 6 2>&1 | grep -v amdgpu.ids | head -2
SNCAL_TT_TRACE=/tmp/tt8.bin timeout 120 python tools/dev/tt_trace_run.py fp16x3 8 > /dev/null 2>&1; python tools/tt_trace.py /tmp/tt8.bin 2>&1 | tail -12; python tools/tt_finish.py /tmp/tt8.bin 2>&1 | tail -10
