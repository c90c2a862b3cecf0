#!/usr/bin/env python3
"""C5 tolerance sweep with enough frames to rank its rows (VERDICT r3 task 7): HRNet-W48 at 1920x1080, deep-path workload, N frames
(default 512, >= 10k usable keypoints) per layer selection of the e4m3 engine, every row against the exact-fp32 engine on the same
frames: moved usable keypoints and cameras beyond 1e-4 relative reprojection error, each with a Wilson 95 % interval, and the wide
convolutions' time.  Rows: the fp32-class engine (fp16x3), the bf16 engine (= selection 'none') and five e4m3 selections.
usage (GPU box): python tools/fp8_sweep_large.py [frames=512] -> gpurun_out/fp8_sweep_large_<frames>.json"""
import json, math, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sncal_amd
import bench

N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
SB = 32
SPECS = ['none', 'c384', 'stage4', 'stage3,stage4', 'c192,c384', 'all']
dev = torch.device('cuda:0')
sd = sncal_amd.synth.peaked_state_dict(bench.seeded_weights('hrnet_w48', seed=1), deep=True)
cc = sncal_amd.CameraCreator(sncal_amd.PITCH_POINTS, **bench.SOLVER_KW)


def wilson(k, n, z=1.96):
    if n == 0:
        return [0.0, 0.0]
    p = k / n
    d = 1 + z * z / n
    c = p + z * z / (2 * n)
    h = z * math.sqrt(p * (1 - p) / n + z * z / (4 * n * n))
    return [round((c - h) / d, 6), round((c + h) / d, 6)]


def batches():
    for lo in range(0, N, SB):
        n = min(SB, N - lo)
        frames, _ = sncal_amd.synth.stamped_frames(n, seed=7000 + lo, size=(1080, 1920))
        yield torch.from_numpy(frames).to(dev)


x3 = sncal_amd._lib.lib().sncal_x3_name().decode()
engines = [('fp32', 'fp32', None), (x3, x3, None)] + [(f'e4m3:{s}' if s != 'none' else 'bf16 (e4m3: none)', 'fp8', s) for s in SPECS]
kps, wide_ms = {}, {}
for name, dtype, spec in engines:
    net = sncal_amd.HRNetHeatmap('hrnet_w48', dtype=dtype, device=dev)
    net.load_state_dict(sd)
    out, ms = [], 0.0
    for bi, x in enumerate(batches()):
        if dtype == 'fp8' and bi == 0:
            net.calibrate_fp8(x[:8])
            net.set_fp8_layers(spec)
        net.set_profiling(True)
        _, kp = net.forward(x, want_heat=False, decode_size=(540, 960))
        prof = {p['kernel']: p for p in net.get_profile()}
        net.set_profiling(False)
        ms += sum(p['ms'] for k, p in prof.items() if k.startswith('conv_tt<'))
        out.append(kp.clone())
        del x
    kps[name] = torch.cat(out, 0)
    wide_ms[name] = ms
    del net
    torch.cuda.empty_cache()
    print(name, 'done', flush=True)
recs = {n: cc.records(cc.solve_device(k)) for n, k in kps.items()}
k32, r32 = kps['fp32'].cpu().numpy(), recs['fp32']
usable = k32[..., 2] >= 0.2
rows = []
for name, _, _ in engines[1:]:
    k, r = kps[name].cpu().numpy(), recs[name]
    moved = (k[..., :2] != k32[..., :2]).any(-1) & usable
    both = [i for i in range(N) if r32[i].status != 0 and r[i].status != 0 and r32[i].rmse > 0]
    far = sum(abs(r[i].rmse - r32[i].rmse) / r32[i].rmse > 1e-4 for i in both)
    rows.append({'engine': name, 'usable_keypoints': int(usable.sum()), 'moved_usable_keypoints': int(moved.sum()),
                 'moved_fraction': round(float(moved.sum()) / max(int(usable.sum()), 1), 6), 'moved_fraction_ci95': wilson(int(moved.sum()), int(usable.sum())),
                 'moved_max_px': float(np.abs(k[..., :2] - k32[..., :2]).max(-1)[usable].max()),
                 'cameras_both': len(both), 'cameras_rmse_beyond_1e-4': int(far), 'cameras_beyond_fraction': round(far / max(len(both), 1), 6),
                 'cameras_beyond_ci95': wilson(int(far), len(both)),
                 'none_ness_changes': sum((r32[i].status == 0) != (r[i].status == 0) for i in range(N)),
                 'wide_3x3_conv_ms_per_64_frames': round(wide_ms[name] / N * 64, 2)})
    print(json.dumps(rows[-1]), flush=True)
os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
json.dump({'workload': f'HRNet-W48 1920x1080, deep-path workload (synth.deep_state_dict), {N} frames, every row against the exact-fp32 engine on the same frames; '
                       'Wilson 95 % intervals; cameras: iterative_voter with the bench solver settings', 'rows': rows},
          open(os.path.join(ROOT, 'gpurun_out', f'fp8_sweep_large_{N}.json'), 'w'), indent=1)
