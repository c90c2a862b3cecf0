"""One fp16x3 W48 forward at the bench size with SNCAL_HEAD_TRACE set (run on the GPU box), then tools/head_trace.py."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault('SNCAL_HEAD_TRACE', 'gpurun_out/head_trace.bin')
import sncal_amd
from bench import seeded_weights
dev = torch.device('cuda:0')
net = sncal_amd.HRNetHeatmap('hrnet_w48', dtype='fp16x3', device=dev)
net.load_state_dict(seeded_weights('hrnet_w48', 1))
x = torch.rand((int(sys.argv[1]) if len(sys.argv) > 1 else 64, 3, 540, 960), device=dev)
for _ in range(2):
    net.forward(x, want_heat=False, decode_size=(540, 960))
torch.cuda.synchronize()
print('trace written')
