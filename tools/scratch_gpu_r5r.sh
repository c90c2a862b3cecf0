#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD; mkdir -p gpurun_out; export TMPDIR=/tmp
SNCAL_BENCH_DIAG=nosolve timeout 600 python bench.py --steps 20 --warmup 5 2>&1 | grep diag
SNCAL_BENCH_DIAG=noprof timeout 600 python bench.py --steps 20 --warmup 5 2>&1 | grep diag
SNCAL_BENCH_DIAG=noprof SNCAL_SOLVE_CUS_PER_XCD=0 timeout 600 python bench.py --steps 20 --warmup 5 2>&1 | grep diag
SNCAL_BENCH_DIAG=noprof SNCAL_SOLVE_STREAMS=2 timeout 600 python bench.py --steps 20 --warmup 5 2>&1 | grep diag
timeout 600 python tools/noisy_pipeline.py 2048 gpurun_out/r5r_noisy_masked.json 2>&1 | grep frames_noisy
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r5r_bench.json 2> gpurun_out/r5r_bench.err; echo "bench rc=$?"; tail -3 gpurun_out/r5r_bench.err
python - <<'PY'
import json
d = json.load(open('gpurun_out/r5r_bench.json'))
print('bench', d['value'], d['ms_per_step'], json.dumps(d['config']['solver']), d['roofline']['avg_launch_us'], d['roofline']['frac'])
print('parity', d.get('parity', {}).get('index_agreement'), d.get('parity', {}).get('frames_rmse_rel_delta_le_1e-4'), d.get('parity', {}).get('cameras_both'))
print('fp32', d.get('fp32', {}).get('value'), d.get('fp32', {}).get('steps'), 'lanes2', d.get('lanes2', {}).get('value'), 'bf16', d.get('bf16', {}).get('value'))
PY
