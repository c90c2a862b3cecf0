"""what does "hidden -> logits in higher precision" buy?  bf16 engine with the head's stage 2 multiplying the
hidden vector as bf16 (default) or as bf16 hi + lo (SNCAL_HEAD_HILO=1, 16 mantissa bits) against the exact-fp32 engine, W48 960x540,
peaked heatmaps (synth.py) at three sharpness settings.  Run once per setting of the variable (it is read at the first launch):
    SNCAL_HEAD_HILO=0 python tools/dev/hilo_experiment.py out0.json;  SNCAL_HEAD_HILO=1 python tools/dev/hilo_experiment.py out1.json"""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import sncal_amd
from bench import seeded_weights
dev = torch.device('cuda:0')
rows = []
for peak, noise in ((12.0, 0.25), (8.0, 0.5), (5.0, 1.0)):
    sd = sncal_amd.synth.peaked_state_dict(seeded_weights('hrnet_w48', 1), peak_logit=peak, noise_gain=noise)
    frames, _ = sncal_amd.synth.stamped_frames(8, seed=4242)
    x = torch.from_numpy(frames).to(dev)
    out = {}
    for dt in ('fp32', 'bf16'):
        net = sncal_amd.HRNetHeatmap('hrnet_w48', dtype=dt, device=dev)
        net.load_state_dict(sd)
        heat, kp = net.forward(x, want_heat=True, decode_size=(540, 960))
        out[dt] = (heat.cpu().numpy(), kp.cpu().numpy())
        del net
    h32, k32 = out['fp32']; h16, k16 = out['bf16']
    d = np.abs(h32 - h16)[:, :57]
    top = h32[:, :57] > np.log(0.01)                      # pixels that carry probability mass
    same = (k32[..., :2] == k16[..., :2]).all(-1)
    usable = k32[..., 2] >= 0.2
    rows.append({'peak_logit': peak, 'noise_gain': noise, 'hilo': os.environ.get('SNCAL_HEAD_HILO', '0'),
                 'abs_dlogp_max': float(d.max()), 'abs_dlogp_mean': float(d.mean()),
                 'abs_dlogp_max_where_p_above_0.01': float(d[top].max()) if top.any() else 0.0,
                 'abs_dlogp_mean_where_p_above_0.01': float(d[top].mean()) if top.any() else 0.0,
                 'index_agreement_all_rows': float(same.mean()), 'index_agreement_usable': float(same[usable].mean()) if usable.any() else 1.0,
                 'conf_abs_delta_max_usable': float(np.abs(k32[..., 2] - k16[..., 2])[usable].max()) if usable.any() else 0.0})
    print(rows[-1], flush=True)
json.dump(rows, open(sys.argv[1], 'w'), indent=1)
