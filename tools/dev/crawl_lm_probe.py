"""One crawling refine_camera fit (tools/scratch/crawl_case.npz: bench frame 16, threshold 0.35, f = 53 px, 8 points; captured from the
oracle by /tmp/capture_crawl.py) through sncal_pnp_refine_lm at the reference's criterion: us per iteration on one wavefront, and the
result against the oracle's (tools/scratch/crawl_case_oracle_result.npz).  GPU box."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import sncal_amd
from sncal_amd import _lib
L = _lib.lib(); dev = torch.device('cuda:0')
c = np.load(os.path.join(ROOT, 'tools', 'scratch', 'crawl_case.npz'))
o = np.load(os.path.join(ROOT, 'tools', 'scratch', 'crawl_case_oracle_result.npz'))
R, t, K4, X, uv = c['R'], c['t'], c['K4'], c['X'], c['uv']
n = len(X)
rt0 = np.concatenate([R.reshape(9), -(R.T @ t)])
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
def run(iters, eps=1e-5):
    d_K = torch.from_numpy(np.tile(K4, (B, 1))).to(dev); d_o = torch.from_numpy(np.tile(X, (B, 1, 1))).to(dev); d_i = torch.from_numpy(np.tile(uv, (B, 1, 1))).to(dev)
    d_n = torch.full((B,), n, dtype=torch.int32, device=dev)
    d_rt = torch.from_numpy(np.tile(rt0, (B, 1))).to(dev); d_rm = torch.zeros((B,), dtype=torch.float64, device=dev)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    _lib.check(L.sncal_pnp_refine_lm(d_K.data_ptr(), d_o.data_ptr(), d_i.data_ptr(), d_n.data_ptr(), B, n, d_rt.data_ptr(), d_rm.data_ptr(), iters, eps, _lib.current_stream_ptr()), 'lm')
    torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3, float(d_rm[0]), d_rt[0].cpu().numpy()
run(10)
for it in (2000, 20000):
    ms, rm, rt = run(it)
    Rg = rt[:9].reshape(3, 3); tg = -(Rg @ rt[9:])
    print(f'{n} points, {it} iterations: {ms:.2f} ms = {ms / it * 1e3:.2f} us per iteration; rmse {rm:.9f}; |R - R_oracle| {np.abs(Rg - o["R"]).max():.3e} |t - t_oracle| {np.abs(tg - o["t"]).max():.3e}')
print('result hash', hash(rt.tobytes()) & 0xffffffff, 'rt', np.array2string(rt, precision=12))
