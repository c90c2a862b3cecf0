#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6e; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -6 $O/pytest.log
python bench.py --workload c4 --steps 20 --warmup 5 --no-cpu-baseline --no-parity > $O/c4_designed.json 2> $O/c4_designed.err
SNCAL_SOLVE_STREAMS=2 python bench.py --workload c4 --steps 20 --warmup 5 --no-cpu-baseline --no-parity > $O/c4_designed_2streams.json 2> $O/c4_designed_2streams.err
SNCAL_BENCH_FORCE_DIST=1 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29513 python bench.py --workload c4 --steps 20 --warmup 5 --no-cpu-baseline --no-parity > $O/c4_designed_dist.json 2> $O/c4_designed_dist.err
python bench.py --workload c4 --line-workload random --steps 10 --warmup 3 --no-cpu-baseline --no-parity > $O/c4_random.json 2> $O/c4_random.err
for f in c4_designed c4_designed_2streams c4_designed_dist c4_random; do python - $O/$f.json <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); s=d['config']['solver']
    print(sys.argv[1].split('/')[-1], d['value'], d['ms_per_step'], 'steady', s['steady_state_ms_per_step'], 'drain', s['drain_ms'], 'nosolve', s['nosolve_ms_per_step'], 'streams', s['solve_streams'], 'cams', d['config']['cameras_found'])
except Exception as e: print(sys.argv[1], 'ERR', e)
PY
done
tail -n 3 $O/*.err | tail -20
