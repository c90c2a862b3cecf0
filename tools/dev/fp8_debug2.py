import os, sys, torch, numpy as np, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ['SNCAL_FP8_DEBUG'] = '1'
import sncal_amd, bench
dev = torch.device('cuda:0')
sd = sncal_amd.synth.peaked_state_dict(bench.seeded_weights('hrnet_w48', 1))
fr, _ = sncal_amd.synth.stamped_frames(2, seed=5, size=(270, 480))
x = torch.from_numpy(fr).to(dev)
n8 = sncal_amd.HRNetHeatmap('hrnet_w48', dtype='fp8', device=dev); n8.load_state_dict(sd)
n8.calibrate_fp8(x)
n8.set_fp8_layers(sys.argv[1] if len(sys.argv) > 1 else 'c96')
for i in range(4):
    print('=== forward', i, file=sys.stderr, flush=True)
    h8, _ = n8.forward(x, want_heat=True)
    torch.cuda.synchronize()
