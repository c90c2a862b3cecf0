#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python tools/dev/solve_streams_probe.py 1024 1 2 4 8 2>&1 | grep "P="
echo "--- trace P=4"
cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_ss -- python $OLDPWD/tools/dev/solve_streams_probe.py 512 4 2>&1 | grep "P="
cd $OLDPWD
f=$(find /tmp/prof_ss -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ks = [r for r in rows if 'voter' in r['Kernel_Name'] or 'calibrate' in r['Kernel_Name']]
t0 = min(int(r['Start_Timestamp']) for r in ks)
ks.sort(key=lambda r: int(r['Start_Timestamp']))
print('n kernels', len(ks), 'columns', list(rows[0].keys())[:14])
for r in ks[:60]:
    print(r['Kernel_Name'][:28], r.get('Queue_Id'), r.get('Stream_Id'), f"{(int(r['Start_Timestamp'])-t0)/1e6:9.2f} -> {(int(r['End_Timestamp'])-t0)/1e6:9.2f} ms", r.get('Scratch_Size'), r.get('VGPR_Count'), r.get('Grid_Size'))
PY
