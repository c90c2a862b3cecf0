#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD; V=r05v28; O=$R/gpurun_out/$V; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_c3.json 2> $O/bench_c3.err; echo "bench rc=$?"
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o trace -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity > $O/bench_c3_traced.json 2> $O/bench_c3_traced.err
find $O/prof -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
find $O/prof -name "*.csv" -size +2M -delete
cd $R
timeout 900 python tools/noisy_pipeline.py 2048 $O/noisy_pipeline_2048.json 2>&1 | grep frames_noisy
python - <<'PY'
import json, os
O = os.path.join(os.environ.get('GRAFT_REPO_ROOT', '/root/repo'), 'gpurun_out', 'r05v28')
d = json.load(open(O + '/bench_c3.json'))
print('bench', d['value'], d['ms_per_step'], json.dumps(d['config']['solver']))
print('roofline', json.dumps(d['roofline']))
print('parity', d.get('parity', {}).get('index_agreement'), d.get('parity', {}).get('frames_rmse_rel_delta_le_1e-4'), d.get('parity', {}).get('cameras_both'))
print('fp32', d.get('fp32', {}).get('value'), 'lanes2', d.get('lanes2', {}).get('value'), 'bf16', d.get('bf16', {}).get('value'))
print('input_stage', json.dumps(d.get('input_stage')))
print('cpu', d.get('cpu_baseline', {}).get('value'), d.get('cpu_baseline', {}).get('stages_fps'))
PY
head -6 $O/kernel_stats.csv
export PMC_B=64 PMC_DTYPE=fp16x3
bash tools/pmc_pass.sh ${V}_x3 FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" 2>&1 | tail -6
cd $R
python tools/pmc_traffic.py gpurun_out/${V}_x3 $V 64 > gpurun_out/${V}_x3/traffic.md 2>&1; tail -24 gpurun_out/${V}_x3/traffic.md
python tools/pmc_mfma.py gpurun_out/${V}_x3 ${V}_fp16x3 2>&1 | tail -14
cp profiles/pmc_traffic.json profiles/${V}_pmc_hbm_traffic.md profiles/${V}_fp16x3_pmc_mfma_util.* gpurun_out/${V}_x3/ 2>/dev/null
