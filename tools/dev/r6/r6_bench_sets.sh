#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6i; mkdir -p $O; cd $R
python bench.py --steps 20 --warmup 5 > $O/bench_c3_sets4.json 2> $O/bench_c3_sets4.err; echo "rc=$?"
python bench.py --steps 20 --warmup 5 --frame-sets 1 --no-cpu-baseline --no-parity > $O/bench_c3_sets1.json 2> $O/bench_c3_sets1.err; echo "rc=$?"
python bench.py --workload c4 --steps 20 --warmup 5 --no-cpu-baseline --no-parity > $O/bench_c4_sets4.json 2> $O/bench_c4_sets4.err; echo "rc=$?"
timeout 600 python -m pytest tests/test_dist_gpu.py -x -q 2>&1 | tail -2
for f in bench_c3_sets4 bench_c3_sets1 bench_c4_sets4; do python - $O/$f.json <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); c=d['config']; s=c['solver']
print(sys.argv[1].split('/')[-1], d['value'], d['ms_per_step'], 'steady', s['steady_state_ms_per_step'], 'drain', s['drain_ms'], 'cams', c['cameras_found'], c['frame_sets']['by_set'], d['roofline']['frac'], (d.get('parity') or {}).get('index_agreement'), (d.get('parity') or {}).get('cameras_both'), (d.get('cpu_baseline') or {}).get('value'))
PY
done
