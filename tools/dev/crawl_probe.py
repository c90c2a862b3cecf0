"""What does the crawling tail of ONE real camera solve cost the network?  The bench's own keypoints (tools/scratch/bench_kp.npy: 164 ms
per batch at the reference's refine criterion, ~8 wavefronts crawl for all of it) are solved on a side stream (plain / CU-masked);
20 ms later -- first pass and the short tasks are over -- ONE forward of 64 frames is timed beside the tail.  GPU box."""
import ctypes, os, sys, time
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch, bench, sncal_amd
from sncal_amd import _lib
dev = torch.device('cuda:0')
NS = torch.cuda.Stream()
sd = sncal_amd.synth.peaked_state_dict(bench.seeded_weights('hrnet_w48', seed=1), deep=True)
net = sncal_amd.HRNetHeatmap('hrnet_w48', dtype='fp16x3', device=dev); net.load_state_dict(sd)
frames, _ = sncal_amd.synth.stamped_frames(64, seed=1000, size=(540, 960))
x = torch.from_numpy(frames).to(dev)
kp = torch.from_numpy(np.load(os.path.join(ROOT, 'tools', 'scratch', 'bench_kp.npy'))).to(dev)
cc = sncal_amd.CameraCreator(sncal_amd.PITCH_POINTS, **bench.SOLVER_KW)
cc200 = sncal_amd.CameraCreator(sncal_amd.PITCH_POINTS, **dict(bench.SOLVER_KW, refine_max_iters=200))
def masked(n):
    h = _lib.vp(); _lib.check(_lib.lib().sncal_stream_create_cu_mask(n, h), 'mask'); return torch.cuda.ExternalStream(h.value)
streams = {'plain stream': torch.cuda.Stream(), 'masked 1 CU per XCD': masked(1), 'masked 2 CUs per XCD': masked(2)}
def fwd_ms(n=1):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.cuda.stream(NS):
        for _ in range(n): net.forward(x, want_heat=False, decode_size=(540, 960))
    NS.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
with torch.cuda.stream(NS):
    for _ in range(3): net.forward(x, want_heat=False, decode_size=(540, 960))
torch.cuda.synchronize()
print('forward alone:', ' '.join(f'{fwd_ms():.2f}' for _ in range(4)))
for name, st in streams.items():
    for what, c in (('20000', cc), ('200', cc200)):
        for rep in range(3):
            torch.cuda.synchronize()
            with torch.cuda.stream(st):
                t0 = time.perf_counter()
                c.solve_device(kp)
            time.sleep(0.02)
            t1 = time.perf_counter()
            with torch.cuda.stream(NS):
                net.forward(x, want_heat=False, decode_size=(540, 960))
            NS.synchronize()
            t2 = time.perf_counter()
            st.synchronize()
            t3 = time.perf_counter()
            print(f'{name}, refine cap {what}: forward beside the tail {(t2 - t1) * 1e3:.2f} ms; solve total {(t3 - t0) * 1e3:.1f} ms', flush=True)
print('forward alone:', ' '.join(f'{fwd_ms():.2f}' for _ in range(4)))
