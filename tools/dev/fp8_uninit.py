import os, sys, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import sncal_amd, bench
dev = torch.device('cuda:0')
sd = sncal_amd.synth.peaked_state_dict(bench.seeded_weights('hrnet_w48', 1))
fr, _ = sncal_amd.synth.stamped_frames(2, seed=5, size=(270, 480))
x = torch.from_numpy(fr).to(dev)
n8 = sncal_amd.HRNetHeatmap('hrnet_w48', dtype=sys.argv[2] if len(sys.argv) > 2 else 'fp8', device=dev); n8.load_state_dict(sd)
if n8.dtype == 2:
    n8.calibrate_fp8(x)
    n8.set_fp8_layers(sys.argv[1] if len(sys.argv) > 1 else 'c96')
res = []
for fill in (0, 255, 127, 0x7e):
    ws = n8._workspace(2, 270, 480)
    ws.fill_(fill)
    h8, _ = n8.forward(x, want_heat=True)
    torch.cuda.synchronize()
    res.append(h8.clone())
    print('fill', fill, 'finite', bool(torch.isfinite(h8).all()), 'equal to fill 0:', bool(torch.equal(h8, res[0])), 'max|d|', float((h8 - res[0]).abs().max()))
