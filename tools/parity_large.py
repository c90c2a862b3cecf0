"""Parity of the fast engines against the exact-fp32 engine on MANY frames of the deep-path workload (bench.py's `parity` object is the
same comparison on the 64 benchmarked frames): keypoint indices and the solved cameras' reprojection error.
usage (GPU box): python tools/parity_large.py [frames=512] [row_gain=0.1] -> gpurun_out/parity_large_<frames>.json"""
import json, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sncal_amd
import bench

N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
row_gain = float(sys.argv[2]) if len(sys.argv) > 2 else 0.1
dev = torch.device('cuda:0')
sd = sncal_amd.synth.peaked_state_dict(bench.seeded_weights('hrnet_w48', seed=1), deep=True, row_gain=row_gain)
# PARITY_TRAINED_LIKE=<seed>: the same function with a trained checkpoint's spread of scales -- every block-internal channel x 2^k, k ~ N(0, 3)
# (four decades), 2 % near-dead channels at 2^-12, the next convolution's column x 2^-k (synth.rescaled_state_dict: bit-identical in fp32).
# PARITY_NO_EQUALIZE=1 switches the load-time rebalancing of the fp16x3 engine off (sncal_hrnet_set_equalize(net, 0)): what it is there for.
trained_like = os.environ.get('PARITY_TRAINED_LIKE')
if trained_like is not None:
    units = sncal_amd.HRNetHeatmap('hrnet_w48', dtype='fp32', device='cpu').conv_units()
    sd = sncal_amd.synth.rescaled_state_dict(sd, units, seed=int(trained_like), sigma_log2=3.0, dead_frac=0.02)
NO_EQUALIZE = os.environ.get('PARITY_NO_EQUALIZE') == '1'
cc = sncal_amd.CameraCreator(sncal_amd.PITCH_POINTS, **bench.SOLVER_KW)
res = {}
kps = {}
for dtype in ('fp32', 'fp16x3', 'bf16'):
    net = sncal_amd.HRNetHeatmap('hrnet_w48', dtype=dtype, device=dev)
    net.equalize = not NO_EQUALIZE
    try:
        net.load_state_dict(sd)
    except sncal_amd._lib.SncalRangeError as e:
        print(dtype, 'REFUSED:', str(e)[:200]); res[dtype] = {'refused': str(e)}; continue
    out = []
    for lo in range(0, N, 64):
        n = min(64, N - lo)
        frames, _ = sncal_amd.synth.stamped_frames(n, seed=5000 + lo, size=(540, 960))
        _, kp = net.forward(torch.from_numpy(frames).to(dev), want_heat=False, decode_size=(540, 960))
        out.append(kp.clone())
    kps[dtype] = torch.cat(out, 0)
    if dtype == 'fp16x3':
        res['fp16x3_range_status'] = list(net.range_status()); res['fp16x3_channels_rebalanced'] = int(getattr(net, 'equalized', 0))
        print('fp16x3 range status', res['fp16x3_range_status'], 'channels rebalanced at load', res['fp16x3_channels_rebalanced'])
    del net
recs = {d: cc.records(cc.solve_device(kps[d])) for d in kps}
for d in ('fp16x3', 'bf16'):
    if d not in kps:
        continue
    res[d] = bench.parity_of(kps['fp32'].cpu().numpy(), recs['fp32'], kps[d].cpu().numpy(), recs[d], 'exact-fp32 engine of this build')
    res[d]['row_gain'] = row_gain
    print(d, {k: res[d][k] for k in ('frames', 'usable_keypoints', 'moved_usable_keypoints', 'index_agreement', 'index_agreement_all_rows', 'cameras_both',
                                     'frames_rmse_rel_delta_le_1e-4', 'rmse_rel_delta_max', 'conf_abs_delta_max_usable', 'conf_signed_delta_mean_usable',
                                     'threshold_crossings')})
# the moved keypoints of fp16x3: how close was the decision in the exact-fp32 heatmap?  (top-1 minus the value at the cell fp16x3 chose, log-probability)
k32, k3 = kps['fp32'].cpu().numpy(), kps['fp16x3'].cpu().numpy()
moved = np.argwhere((k32[..., 2] >= 0.2) & ((k32[..., :2] != k3[..., :2]).any(-1)))
net = sncal_amd.HRNetHeatmap('hrnet_w48', dtype='fp32', device=dev)
net.load_state_dict(sd)
gaps = []
for f, k in moved[:32]:
    lo = int(f) // 64 * 64
    n = min(64, N - lo)
    frames, _ = sncal_amd.synth.stamped_frames(n, seed=5000 + lo, size=(540, 960))
    heat, _ = net.forward(torch.from_numpy(frames[int(f) - lo:int(f) - lo + 1]).to(dev), want_heat=True, decode_size=(540, 960))
    h = heat[0, int(k)].cpu().numpy()                      # (270, 480) log-probabilities
    sx, sy = 960 / h.shape[1], 540 / h.shape[0]
    c32 = (int(round(k32[f, k, 1] / sy)), int(round(k32[f, k, 0] / sx)))
    c3 = (int(round(k3[f, k, 1] / sy)), int(round(k3[f, k, 0] / sx)))
    c32 = (min(c32[0], h.shape[0] - 1), min(c32[1], h.shape[1] - 1)); c3 = (min(c3[0], h.shape[0] - 1), min(c3[1], h.shape[1] - 1))
    gaps.append(dict(frame=int(f), keypoint=int(k), fp32_cell=c32, fp16x3_cell=c3, fp32_logp_at_fp32_cell=float(h[c32]), fp32_logp_at_fp16x3_cell=float(h[c3]),
                     gap=float(h.max() - h[c3]), conf=float(k32[f, k, 2])))
    print('moved', gaps[-1])
res['fp16x3']['moved_keypoints_fp32_gap'] = gaps
# frames whose cameras differ although no usable index moved: does a confidence cross one of the solver's thresholds?
r32, r3 = recs['fp32'], recs['fp16x3']
odd = []
for i in range(N):
    if r32[i].status == 0 or r3[i].status == 0 or r32[i].rmse <= 0:
        continue
    if abs(r3[i].rmse - r32[i].rmse) / r32[i].rmse <= 1e-4:
        continue
    usable = k32[i, :, 2] >= 0.2
    if ((k32[i, :, :2] != k3[i, :, :2]).any(-1) & usable).any():
        continue
    cross = [(int(k), float(k32[i, k, 2]), float(k3[i, k, 2])) for k in range(k32.shape[1])
             for th in (0.5, 0.35, 0.2) if (k32[i, k, 2] >= th) != (k3[i, k, 2] >= th)]
    other = [(int(k), float(k32[i, k, 0]), float(k32[i, k, 1]), float(k3[i, k, 0]), float(k3[i, k, 1])) for k in range(k32.shape[1])
             if (k32[i, k, :2] != k3[i, k, :2]).any()]
    odd.append(dict(frame=i, rmse_fp32=float(r32[i].rmse), rmse_fp16x3=float(r3[i].rmse), threshold_crossings=cross, moved_unusable_rows=other[:4]))
    print('camera differs, no usable index moved:', odd[-1])
res['fp16x3']['cameras_differ_without_moved_usable_index'] = odd
os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
res['split_type'] = sncal_amd._lib.lib().sncal_x3_name().decode()
res['trained_like_seed'] = trained_like
tag = os.environ.get('PARITY_TAG', res['split_type'] + ('' if trained_like is None else f'_trainedlike{trained_like}') + ('_noeq' if os.environ.get('PARITY_NO_EQUALIZE') == '1' else ''))
json.dump(res, open(os.path.join(ROOT, 'gpurun_out', f'parity_large_{N}_rowgain{row_gain}_{tag}.json'), 'w'), indent=1)
