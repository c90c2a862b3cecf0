"""One worker of the CPU baseline's solve pool (bench.py cpu_baseline leg): TEST / MEASUREMENT INFRASTRUCTURE ONLY.

    python oracle/solve_worker.py <kpts.npy> <start> <count> [refine_max_iters]

Mirrors one of the reference's 16 `ProcessPoolExecutor` workers (/root/reference/src/utils/make_submit.py:25,53-54):
solves `count` frames of the (n,57,3) keypoint file with the oracle CameraCreator and prints
"<seconds> <cameras found>" for the solves alone (interpreter start-up and imports excluded; one untimed warm-up call).
A plain script started with subprocess, so that no worker ever imports torch or touches the GPU runtime.
"""
import contextlib
import io
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import solve as osolve  # noqa: E402


def main():
    kp = np.load(sys.argv[1])
    start, count = int(sys.argv[2]), int(sys.argv[3])
    if len(sys.argv) > 4:                      # the bench's explicit cap on refine_camera's LM (bench.py REFINE_CAP): like for like with the GPU leg
        osolve.opencv_stops(int(sys.argv[4]))
    oc = osolve.CameraCreatorOracle()
    found = 0
    with contextlib.redirect_stdout(io.StringIO()):
        oc(kp[start % len(kp)], None)
        t0 = time.perf_counter()
        for i in range(count):
            found += oc(kp[(start + i) % len(kp)], None) is not None
        dt = time.perf_counter() - t0
    print(f'{dt:.6f} {found}')


if __name__ == '__main__':
    main()
