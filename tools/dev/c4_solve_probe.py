"""C4's solve tail: the bench's C4 frames once through both networks, then the solve of THOSE keypoints + line points
timed synchronously on an unmasked stream and on CU-masked streams of 1 / 2 / 4 CUs per XCD, with and without the line points.
Run under `rocprofv3 --kernel-trace --stats` for the per-kernel split (first_pass_task / voter_task ...)."""
import os, sys, time
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import sncal_amd, bench
from sncal_amd.lines import lines_to_points_device
from sncal_amd.transforms import EHMPredictionTransform
dev = torch.device('cuda:0')
torch.cuda.set_stream(torch.cuda.Stream(device=dev))
B = 64
sd = sncal_amd.synth.peaked_state_dict(bench.seeded_weights('hrnet_w48', seed=1), deep=True)
net = sncal_amd.HRNetHeatmap('hrnet_w48', dtype='fp16x3', device=dev); net.load_state_dict(sd)
ln = sncal_amd.HRNetHeatmap('line_hrnet_w48', dtype='fp16x3', device=dev); ln.load_state_dict(bench.seeded_weights('line_hrnet_w48', seed=2))
frames, _ = sncal_amd.synth.stamped_frames(B, seed=1000, size=(540, 960))
x = torch.from_numpy(frames).to(dev)
_, kpts = net.forward(x, want_heat=False, decode_size=(540, 960))
heat, _ = ln.forward(x, want_heat=True)
peaks = EHMPredictionTransform.mask_heat_points_gauss(heat, sigma=3.0)
lp = lines_to_points_device(peaks, scale=4.0, prob_thre=0.0)
torch.cuda.synchronize()
lpn = lp.cpu().numpy(); kn = kpts.cpu().numpy()
print('valid line points per frame: mean %.1f min %d max %d; in image: %.1f' % ((lpn[..., 2] > 0.5).sum(1).mean(), (lpn[..., 2] > 0.5).sum(1).min(), (lpn[..., 2] > 0.5).sum(1).max(),
      ((lpn[..., 2] > 0.5) & (lpn[..., 0] >= 0) & (lpn[..., 0] <= 960) & (lpn[..., 1] >= 0) & (lpn[..., 1] <= 540)).sum(1).mean()))
print('keypoints above 0.5 per frame: mean %.1f' % (kn[..., 2] > 0.5).sum(1).mean())
np.savez(os.path.join(ROOT, 'gpurun_out', 'c4_solve_inputs.npz'), kpts=kn, line_pts=lpn)
cc = sncal_amd.CameraCreator(sncal_amd.PITCH_POINTS, **bench.SOLVER_KW)

def timed(stream, with_lines, reps=2):
    with torch.cuda.stream(stream):
        out = cc.solve_device(kpts, lp if with_lines else None)
        stream.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            out = cc.solve_device(kpts, lp if with_lines else None)
        stream.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3, out

plain = torch.cuda.Stream(device=dev)
for wl in (False, True):
    ms, out = timed(plain, wl)
    recs = cc.records(out)
    st = np.array([r.status for r in recs])
    print('unmasked stream, lines=%d: %.1f ms; cameras %d/%d, status histogram %s' % (wl, ms, (st != 0).sum(), B, dict(zip(*np.unique(st, return_counts=True)))))
for cus in (1, 2, 4):
    h = sncal_amd._lib.vp()
    sncal_amd._lib.check(sncal_amd._lib.lib().sncal_stream_create_cu_mask(cus, h), 'mask')
    s = torch.cuda.ExternalStream(h.value, device=dev)
    for wl in (False, True):
        ms, _ = timed(s, wl)
        print('masked %d CU/XCD, lines=%d: %.1f ms' % (cus, wl, ms))
