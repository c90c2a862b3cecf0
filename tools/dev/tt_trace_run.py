"""One W48 forward at the bench size with SNCAL_TT_TRACE set (run on the GPU box), then tools/tt_trace.py.
usage: SNCAL_TT_TRACE=gpurun_out/x/tt.bin [SNCAL_TT_TRACE_CFG64=1] python tools/dev/tt_trace_run.py [dtype=fp16x3] [batch=64]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault('SNCAL_TT_TRACE', 'gpurun_out/tt_trace.bin')
os.makedirs(os.path.dirname(os.environ['SNCAL_TT_TRACE']) or '.', exist_ok=True)
import sncal_amd
from bench import seeded_weights
dtype = sys.argv[1] if len(sys.argv) > 1 else 'fp16x3'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
dev = torch.device('cuda:0')
net = sncal_amd.HRNetHeatmap('hrnet_w48', dtype=dtype, device=dev)
net.load_state_dict(seeded_weights('hrnet_w48', 1))
x = torch.rand((B, 3, 540, 960), device=dev)
for _ in range(2):
    net.forward(x, want_heat=False, decode_size=(540, 960))
torch.cuda.synchronize()
print('trace written', os.environ['SNCAL_TT_TRACE'])
