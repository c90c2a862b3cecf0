"""Where the side-stream solve's wall time goes on the bench's own keypoints (GPU box): the whole batch, the frames the first pass
(original_voter) leaves pending, and the voter at each of iterative_voter's thresholds on exactly those frames."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import sncal_amd
import bench

dev = torch.device('cuda:0')
sd = sncal_amd.synth.peaked_state_dict(bench.seeded_weights('hrnet_w48', seed=1), deep=True)
net = sncal_amd.HRNetHeatmap('hrnet_w48', dtype='fp16x3', device=dev)
net.load_state_dict(sd)
frames, _ = sncal_amd.synth.stamped_frames(64, seed=1000, size=(540, 960))
x = torch.from_numpy(frames).to(dev)
_, kp = net.forward(x, want_heat=False, decode_size=(540, 960))
torch.cuda.synchronize()


def timed(cc, k, reps=3):
    out = cc.solve_device(k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        cc.solve_device(k, out=out)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3, cc.records(out)


kw = dict(bench.SOLVER_KW)
cc = sncal_amd.CameraCreator(sncal_amd.PITCH_POINTS, **kw)
ms, recs = timed(cc, kp)
tags = [r.status for r in recs]
print(f'iterative_voter, 64 frames: {ms:.2f} ms; tags {sorted(set(tags))}: ' + ', '.join(f'{t}: {tags.count(t)}' for t in sorted(set(tags))))
kw1 = dict(kw, algorithm='original_voter')
ms1, recs1 = timed(sncal_amd.CameraCreator(sncal_amd.PITCH_POINTS, **kw1), kp)
pend = [i for i, r in enumerate(recs1) if r.status == 0]
print(f'original_voter alone: {ms1:.2f} ms; frames without a camera (pending for the voter): {len(pend)} {pend}')
if pend:
    kpp = kp[pend].contiguous()
    for thr in kw['conf_threshs']:
        kwv = dict(kw, algorithm='voter', conf_thresh=thr)
        for sel in (list(range(len(pend))),) + tuple([i] for i in range(len(pend))):
            msv, rv = timed(sncal_amd.CameraCreator(sncal_amd.PITCH_POINTS, **kwv), kpp[sel].contiguous())
            print(f'  voter at {thr}: frames {[pend[i] for i in sel]}: {msv:.2f} ms (one wave per frame), cameras {[r.status for r in rv]}')
    msi, ri = timed(cc, kpp)
    print(f'iterative_voter on the pending frames only: {msi:.2f} ms, tags {[r.status for r in ri]}')
