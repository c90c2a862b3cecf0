#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD; mkdir -p gpurun_out; export TMPDIR=/tmp
cd /tmp && timeout 900 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/prof_wt -- python $R/tools/dev/solve_wavetime.py 256 2>&1 | grep -v amdgpu.ids | tail -12
cd $R
ls /tmp/prof_wt/*/* | head
python - <<'PY'
import csv, glob, collections
f = glob.glob('/tmp/prof_wt/**/*counter_collection.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
print(list(rows[0].keys()))
d = collections.OrderedDict()
for r in rows:
    k = (r['Dispatch_Id'], r['Kernel_Name'][:40])
    d.setdefault(k, {})[r['Counter_Name']] = float(r['Counter_Value'])
for (i, k), c in d.items():
    if 'calib' in k or 'voter_task' in k:
        gui = c.get('GRBM_GUI_ACTIVE', 0)
        print(i, k[-30:], {n: int(v) for n, v in c.items()}, 'mean resident waves ~', round(c.get('SQ_WAVE_CYCLES', 0) * 4 / max(gui, 1), 1) if gui else None)
PY
