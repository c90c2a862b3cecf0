"""A/B helper: hash of the bf16 W48 network output (heatmap + keypoints) for the current environment; run it under
different tuning env vars (read once per process) and compare the hashes."""
import hashlib, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import sncal_amd
from bench import seeded_weights
dev = torch.device('cuda:0')
net = sncal_amd.HRNetHeatmap('hrnet_w48', dtype='bf16', device=dev)
net.load_state_dict(seeded_weights('hrnet_w48', 1))
x = torch.rand((int(sys.argv[1]) if len(sys.argv) > 1 else 8, 3, 540, 960), device=dev, generator=torch.Generator(device=dev).manual_seed(5))
h = hashlib.sha256()
for _ in range(3):
    heat, kp = net.forward(x, want_heat=True, decode_size=(540, 960))
    h.update(heat.cpu().numpy().tobytes()); h.update(kp.cpu().numpy().tobytes())
print('sha', h.hexdigest()[:20])
