import os, sys, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ['SNCAL_FP8_DEBUG'] = '1'
import sncal_amd, bench
dev = torch.device('cuda:0')
sd = sncal_amd.synth.peaked_state_dict(bench.seeded_weights('hrnet_w48', 1))
fr, _ = sncal_amd.synth.stamped_frames(2, seed=5, size=(270, 480))
x = torch.from_numpy(fr).to(dev)
nb = sncal_amd.HRNetHeatmap('hrnet_w48', dtype='bf16', device=dev); nb.load_state_dict(sd)
h16, _ = nb.forward(x, want_heat=True)
n8 = sncal_amd.HRNetHeatmap('hrnet_w48', dtype='fp8', device=dev); n8.load_state_dict(sd)
n8.calibrate_fp8(x)
for spec in sys.argv[1:] or ['stage2']:
    n8.set_fp8_layers(spec)
    h8, _ = n8.forward(x, want_heat=True)
    d = (h8 - h16).abs()
    print(spec, 'dlogp mean/max', float(d.mean()), float(d.max()), flush=True)
