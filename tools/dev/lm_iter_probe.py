"""Cost of one iteration of refine_camera's LMSolver run on one wavefront (sncal_pnp_refine_lm with an unreachable eps): GPU box."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import sncal_amd
from sncal_amd import _lib
from oracle import synth
from oracle.pitch import pitch_points
L = _lib.lib()
dev = torch.device('cuda:0')
P = pitch_points()
kp, cam = synth.synth_keypoints(7, sigma_px=1.0)
ids = [i for i in range(57) if kp[i, 2] > 0.5]
obj = np.ascontiguousarray(P[ids]); img = np.ascontiguousarray(kp[ids, :2].astype(np.float64))
n = len(ids)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
K = np.tile(np.array([2000.0, 2000.0, 480.0, 270.0]), (B, 1))
rt0 = np.concatenate([np.eye(3).reshape(9), np.array([0.0, 60.0, -20.0])])
def run(iters):
    d_K = torch.from_numpy(K).to(dev); d_o = torch.from_numpy(np.tile(obj, (B, 1, 1))).to(dev); d_i = torch.from_numpy(np.tile(img, (B, 1, 1))).to(dev)
    d_n = torch.full((B,), n, dtype=torch.int32, device=dev)
    d_rt = torch.from_numpy(np.tile(rt0, (B, 1))).to(dev); d_rm = torch.zeros((B,), dtype=torch.float64, device=dev)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    _lib.check(L.sncal_pnp_refine_lm(d_K.data_ptr(), d_o.data_ptr(), d_i.data_ptr(), d_n.data_ptr(), B, n, d_rt.data_ptr(), d_rm.data_ptr(), iters, 1e-300, _lib.current_stream_ptr()), 'lm')
    torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3, float(d_rm[0])
run(10)
for it in (100, 1000, 4000):
    ms, rm = run(it)
    print(f'{n} points, B={B}, {it} iterations: {ms:.2f} ms = {ms / it * 1e3:.2f} us per iteration (rmse {rm:.3f})')
