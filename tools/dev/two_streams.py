import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import sncal_amd
from bench import seeded_weights
dev = torch.device('cuda:0')
sd = seeded_weights('hrnet_w48', 1)
NS = int(sys.argv[1]); B = int(sys.argv[2])
nets = []
for i in range(NS):
    n = sncal_amd.HRNetHeatmap('hrnet_w48', dtype='bf16', device=dev); n.load_state_dict(sd); nets.append(n)
xs = [torch.rand((B, 3, 540, 960), device=dev) for _ in range(NS)]
streams = [torch.cuda.Stream() for _ in range(NS)]
def step():
    for n, x, s in zip(nets, xs, streams):
        with torch.cuda.stream(s):
            n.forward(x, want_heat=False, decode_size=(540, 960))
for _ in range(2): step()
torch.cuda.synchronize()
t0 = time.time(); K = 4
for _ in range(K): step()
torch.cuda.synchronize()
dt = (time.time() - t0) / K
print(f'{NS} streams x B={B}: {dt*1e3:.1f} ms/step, {NS*B/dt:.1f} frames/s')
