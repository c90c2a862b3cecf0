#!/bin/bash
O=gpurun_out/r4j; mkdir -p $O
timeout 900 python -m pytest tests/test_solve_gpu.py -m gpu -x -q > $O/pytest.txt 2>&1; tail -8 $O/pytest.txt
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/r4j/bench.json'))
print({k: d[k] for k in ('value','ms_per_step')}, d['config'].get('solve_ms_per_batch'), d['config'].get('cameras_found'), d['roofline']['kernel'], d['roofline']['frac'])
print(d['parity'])
print({k:(v['share_of_gpu_time'], v['avg_launch_us']) for k,v in d['kernels'].items()})
PY
tail -3 $O/bench.err
