#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
SNCAL_TT_TRACE=/tmp/tt_new.bin python tools/dev/tt_trace_run.py > /dev/null 2>&1; echo "== queue (this build)"; python tools/tt_trace.py /tmp/tt_new.bin 2>&1 | tail -22
SNCAL_LIB_PATH=tools/ab/libsncal_r4.so SNCAL_TT_TRACE=/tmp/tt_r4.bin python tools/dev/tt_trace_run.py > /dev/null 2>&1; echo "== static deal (round 4 build)"; python tools/tt_trace.py /tmp/tt_r4.bin 2>&1 | tail -22
python tools/tt_finish.py /tmp/tt_new.bin 2>&1 | tail -6; python tools/tt_finish.py /tmp/tt_r4.bin 2>&1 | tail -6
