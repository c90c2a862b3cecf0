#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
python tools/dev/solve_tasks_check.py run /tmp/a1.npz 256 > /dev/null 2>&1
python tools/dev/solve_tasks_check.py run /tmp/a2.npz 256 > /dev/null 2>&1
SNCAL_SOLVE_WAVE_WGS=0 python tools/dev/solve_tasks_check.py run /tmp/c.npz 256 > /dev/null 2>&1
SNCAL_SOLVE_WAVE_WGS=0 python tools/dev/solve_tasks_check.py run /tmp/c2.npz 256 > /dev/null 2>&1
python - <<'PY'
import numpy as np, ctypes, sys
sys.path.insert(0, '.')
from sncal_amd import _lib
def recs(a): 
    return (_lib.Camera * (a.shape[0])).from_buffer_copy(a.tobytes())
A1, A2, C, C2 = (np.load(f'/tmp/{n}.npz') for n in ('a1', 'a2', 'c', 'c2'))
for k in ('rec_synth_cap20000', 'rec_synth_cap200'):
    print(k, 'default twice equal', np.array_equal(A1[k], A2[k]), '; paired twice equal', np.array_equal(C[k], C2[k]), '; default vs paired', np.array_equal(A1[k], C[k]))
    d = np.where((A1[k] != C[k]).any(1))[0]
    ra, rc = recs(A1[k]), recs(C[k])
    for i in d[:8]:
        print('  frame', i, 'status', ra[i].status, rc[i].status, 'rmse', ra[i].rmse, rc[i].rmse, 'fx', ra[i].fx, rc[i].fx)
PY
