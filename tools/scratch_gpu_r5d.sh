#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python tools/noisy_pipeline.py 2048 gpurun_out/r5d_noisy_pool4.json 2>&1 | grep frames_noisy
SNCAL_SOLVE_STREAMS=6 timeout 600 python tools/noisy_pipeline.py 2048 gpurun_out/r5d_noisy_pool6.json 2>&1 | grep frames_noisy
SNCAL_LIB_PATH=tools/ab/libsncal_r4.so timeout 600 python tools/noisy_pipeline.py 1024 gpurun_out/r5d_noisy_pool4_r4lib.json 2>&1 | grep frames_noisy
timeout 900 python bench.py --no-cpu-baseline > gpurun_out/r5d_bench.json 2> gpurun_out/r5d_bench.err; echo "bench rc=$?"; tail -3 gpurun_out/r5d_bench.err
python - <<'PY'
import json
d = json.load(open('gpurun_out/r5d_bench.json'))
print('bench', d['value'], d['ms_per_step'], json.dumps(d['config']['solver']), d['roofline']['avg_launch_us'], d['roofline']['frac'])
print('parity', d.get('parity', {}).get('index_agreement'), d.get('parity', {}).get('frames_rmse_rel_delta_le_1e-4'), d.get('parity', {}).get('cameras_both'))
print('fp32', d.get('fp32', {}).get('value'), d.get('fp32', {}).get('steps'), 'lanes2', d.get('lanes2', {}).get('value'))
PY
cd /tmp && timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_np -- python $R/tools/noisy_pipeline.py 512 2>&1 | grep frames_noisy
cd $R
f=$(find /tmp/prof_np -name "*kernel_trace.csv" | head -1)
python tools/dev/trace_queues.py "$f"
