#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD; O=$R/gpurun_out/r05v25; mkdir -p $O; export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_c3.json 2> $O/bench_c3.err; echo "bench rc=$?"
timeout 900 python tools/noisy_pipeline.py 2048 $O/noisy_pipeline_2048.json 2>&1 | grep frames_noisy
python - <<'PY'
import json, os
O = os.path.join(os.environ.get('GRAFT_REPO_ROOT', '/root/repo'), 'gpurun_out', 'r05v25')
d = json.load(open(O + '/bench_c3.json'))
print('bench', d['value'], d['ms_per_step'], json.dumps(d['config']['solver']))
print('roofline', d['roofline']['avg_launch_us'], d['roofline']['frac'], 'solve_ms_per_batch', d['config']['solve_ms_per_batch'])
print('parity', d.get('parity', {}).get('index_agreement'), d.get('parity', {}).get('frames_rmse_rel_delta_le_1e-4'), d.get('parity', {}).get('cameras_both'))
PY
