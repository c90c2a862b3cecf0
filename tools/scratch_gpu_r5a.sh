#!/bin/bash
# round 5, GPU session A: queue-dealt conv_tt + solve-stream pool at the reference's refine criterion
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_pipeline_gpu.py tests/test_kernels_gpu.py tests/test_hrnet_gpu.py tests/test_dist_gpu.py -m gpu -x -q > gpurun_out/r5a_pytest.log 2>&1
echo "pytest rc=$?"; tail -5 gpurun_out/r5a_pytest.log
timeout 600 python tools/noisy_pipeline.py 2048 gpurun_out/r5a_noisy_pool4.json 2>&1 | tail -2
SNCAL_LIB_PATH=tools/ab/libsncal_r4.so timeout 600 python tools/noisy_pipeline.py 1024 gpurun_out/r5a_noisy_pool4_r4lib.json 2>&1 | tail -2
SNCAL_SOLVE_STREAMS=1 timeout 600 python tools/noisy_pipeline.py 512 gpurun_out/r5a_noisy_pool1.json 2>&1 | tail -2
timeout 900 python bench.py > gpurun_out/r5a_bench.json 2> gpurun_out/r5a_bench.err; echo "bench rc=$?"; tail -3 gpurun_out/r5a_bench.err
python - <<'PY'
import json
try:
    d = json.load(open('gpurun_out/r5a_bench.json'))
    print('bench', d['value'], d['ms_per_step'], json.dumps(d['config']['solver']), d['roofline']['avg_launch_us'], d['roofline']['frac'])
    print('parity', d.get('parity', {}).get('index_agreement'), d.get('parity', {}).get('frames_rmse_rel_delta_le_1e-4'), d.get('parity', {}).get('cameras_both'))
    print('fp32', d.get('fp32', {}).get('value'), 'cpu', d.get('cpu_baseline', {}).get('value'))
except Exception as e:
    print('no bench line', e)
PY
