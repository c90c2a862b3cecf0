#!/bin/bash
# round 6, first GPU call: the whole -m gpu suite, then the bench lines the once-at-end gather must not move
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6a; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -5 $O/pytest.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity > $O/c3_plain.json 2> $O/c3_plain.err
SNCAL_BENCH_FORCE_DIST=1 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29511 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity > $O/c3_dist.json 2> $O/c3_dist.err
python bench.py --workload c4 --steps 20 --warmup 5 --no-cpu-baseline --no-parity > $O/c4_plain.json 2> $O/c4_plain.err
SNCAL_BENCH_FORCE_DIST=1 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29512 python bench.py --workload c4 --steps 20 --warmup 5 --no-cpu-baseline --no-parity > $O/c4_dist.json 2> $O/c4_dist.err
for f in c3_plain c3_dist c4_plain c4_dist; do python - $O/$f.json <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); s=d['config']['solver']
    print(sys.argv[1].split('/')[-1], d['value'], d['ms_per_step'], 'steady', s['steady_state_ms_per_step'], 'drain', s['drain_ms'], 'nosolve', s['nosolve_ms_per_step'], 'streams', s['solve_streams'], d['roofline']['frac'], d['roofline'].get('frac_of_split_ceiling'), d['roofline']['avg_launch_us'])
except Exception as e: print(sys.argv[1], 'ERR', e)
PY
done
tail -3 $O/*.err | tail -30
