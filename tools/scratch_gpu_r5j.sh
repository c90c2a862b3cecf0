#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_solve_gpu.py tests/test_pipeline_gpu.py tests/test_configs_gpu.py -m gpu -x -q 2>&1 | tail -4
timeout 600 python tools/noisy_pipeline.py 2048 gpurun_out/r5j_noisy_masked.json 2>&1 | grep frames_noisy
SNCAL_NULL_STREAM=1 timeout 600 python tools/noisy_pipeline.py 1024 gpurun_out/r5j_noisy_nullstream.json 2>&1 | grep frames_noisy
SNCAL_SOLVE_CUS_PER_XCD=2 timeout 600 python tools/noisy_pipeline.py 1024 gpurun_out/r5j_noisy_masked2.json 2>&1 | grep frames_noisy
timeout 900 python bench.py --no-cpu-baseline > gpurun_out/r5j_bench.json 2> gpurun_out/r5j_bench.err; echo "bench rc=$?"; tail -3 gpurun_out/r5j_bench.err
python - <<'PY'
import json
d = json.load(open('gpurun_out/r5j_bench.json'))
print('bench', d['value'], d['ms_per_step'], json.dumps(d['config']['solver']), d['roofline']['avg_launch_us'], d['roofline']['frac'])
print('parity', d.get('parity', {}).get('index_agreement'), d.get('parity', {}).get('frames_rmse_rel_delta_le_1e-4'), d.get('parity', {}).get('cameras_both'))
print('fp32', d.get('fp32', {}).get('value'), d.get('fp32', {}).get('steps'), 'lanes2', d.get('lanes2', {}).get('value'), 'bf16', d.get('bf16', {}).get('value'))
PY
