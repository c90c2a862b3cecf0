"""GPU: the multi-GPU code path of bench.py on ONE GPU.  `SNCAL_BENCH_FORCE_DIST=1` makes a single rank take it: RCCL
process-group init bound to the device, the records of every step logged on the rank, the ONE `all_gather_into_tensor` of all steps'
records behind the last step's solves (CalibrationPipeline.gather_all), the barrier and the max-over-ranks reduction -- everything the 8-GPU driver run relies on except a second rank.
The world-size-2 logic (ragged shards, frame order) is covered on gloo in tests/test_dist.py."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def test_forced_rccl_single_rank_gather_equals_local_records(tmp_path):
    dump = str(tmp_path / 'gather.npz')
    env = dict(os.environ, SNCAL_BENCH_FORCE_DIST='1', RANK='0', LOCAL_RANK='0', WORLD_SIZE='1', MASTER_ADDR='127.0.0.1',
               MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY='0')
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '2', '--warmup', '1', '--batch', '8',
           '--no-cpu-baseline', '--no-parity', '--dump-gather', dump]
    r = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    line = json.loads([ln for ln in r.stdout.decode().splitlines() if ln.startswith('{')][-1])
    assert line['n_gpus'] == 1 and line['steps'] == 2 and line['value'] > 0
    assert 'all_gather' not in line['config']['parallelism'] or line['n_gpus'] > 1          # one rank: labelled single GPU
    d = np.load(dump)
    assert int(d['world']) == 1
    # one rank: the gathered tensor IS the rank's own records of BOTH steps, byte for byte (keypoints 684 B + the camera record per
    # frame), in submission order: the last 8 rows are the last step's
    assert d['gathered'].dtype == np.uint8 and d['gathered'].shape == d['local'].shape and d['local'].shape[0] == 2 * 8
    assert np.array_equal(d['gathered'], d['local'])
    assert np.array_equal(d['local'][8:], d['last_step'])
    assert d['local'].shape[1] > 57 * 3 * 4
    # the collective is not part of a step: all three solve streams stay (rounds 2-5 gave RCCL one of them)
    assert line['config']['solver']['solve_streams'] == 3
