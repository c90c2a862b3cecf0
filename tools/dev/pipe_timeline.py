"""Where does a pipeline step's time go?  Per step: host time inside submit(), GPU time of the forward (events on the network's stream),
GPU idle time between consecutive forwards.  GPU box: python tools/dev/pipe_timeline.py [steps=24]   (SNCAL_BENCH_REFINE_CAP, SNCAL_SOLVE_*)"""
import os, sys, time
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch, bench, sncal_amd
dev = torch.device('cuda:0')
if os.environ.get('SNCAL_NULL_STREAM') != '1':
    torch.cuda.set_stream(torch.cuda.Stream(device=dev))
K = int(sys.argv[1]) if len(sys.argv) > 1 else 24
sd = sncal_amd.synth.peaked_state_dict(bench.seeded_weights('hrnet_w48', seed=1), deep=True)
net = sncal_amd.HRNetHeatmap('hrnet_w48', dtype='fp16x3', device=dev); net.load_state_dict(sd)
frames, _ = sncal_amd.synth.stamped_frames(64, seed=1000, size=(540, 960))
x = torch.from_numpy(frames).to(dev)
cc = sncal_amd.CameraCreator(sncal_amd.PITCH_POINTS, **bench.SOLVER_KW)
pipe = sncal_amd.CalibrationPipeline(net, cc, decode_size=(540, 960))
for _ in range(3): net.forward(x, want_heat=False, decode_size=(540, 960))
torch.cuda.synchronize()
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(K)]
host = []
fwd = net.forward
t_start = time.perf_counter()
for k in range(K):
    e0, e1, e2 = ev[k]
    def timed_forward(*a, **kw):
        e0.record()
        r = fwd(*a, **kw)
        e1.record()
        return r
    net.forward = timed_forward
    t0 = time.perf_counter()
    out = pipe.submit(x)
    host.append((time.perf_counter() - t0) * 1e3)
net.forward = fwd
t_enq = time.perf_counter()
pipe.join(); torch.cuda.synchronize()
t_end = time.perf_counter()
f = [ev[k][0].elapsed_time(ev[k][1]) for k in range(K)]
gap = [ev[k][1].elapsed_time(ev[k + 1][0]) for k in range(K - 1)]
print(f'masked {pipe.masked} cap {cc.refine_max_iters}: wall per step {(t_end - t_start) / K * 1e3:.2f} ms (enqueue loop {(t_enq - t_start) / K * 1e3:.2f} ms per step)')
print('host ms inside submit():', ' '.join(f'{h:.1f}' for h in host))
print('GPU forward ms:        ', ' '.join(f'{v:.1f}' for v in f))
print('GPU gap to next forward:', ' '.join(f'{v:.2f}' for v in gap))
