#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD; mkdir -p gpurun_out; export TMPDIR=/tmp
cd /tmp && timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_np -- python $R/tools/noisy_pipeline.py 1024 2>&1 | grep frames_noisy
cd $R
f=$(find /tmp/prof_np -name "*kernel_trace.csv" | head -1)
python tools/dev/trace_phases.py "$f"
gzip -c "$f" > gpurun_out/r5e_noisy_trace.csv.gz; ls -la gpurun_out/r5e_noisy_trace.csv.gz
