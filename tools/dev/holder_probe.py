"""What does a kernel on ANOTHER stream cost the network?  The fp16x3 W48 forward of 64 frames (K steps) alone, then beside k spinning
single-wave workgroups of several kinds (tools/dev/holder.hip) that outlive the steps.  GPU box: python tools/dev/holder_probe.py"""
import ctypes, os, sys, time
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch, bench, sncal_amd
H = ctypes.CDLL(os.path.join(ROOT, 'tools', 'scratch', 'libholder.so'))
H.holder_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_void_p]
dev = torch.device('cuda:0')
sd = sncal_amd.synth.peaked_state_dict(bench.seeded_weights('hrnet_w48', seed=1), deep=True)
net = sncal_amd.HRNetHeatmap('hrnet_w48', dtype='fp16x3', device=dev); net.load_state_dict(sd)
frames, _ = sncal_amd.synth.stamped_frames(64, seed=1000, size=(540, 960))
x = torch.from_numpy(frames).to(dev)
side = torch.cuda.Stream()
K = 6
def run(variant=None, k=0):
    for _ in range(2): net.forward(x, want_heat=False, decode_size=(540, 960))
    torch.cuda.synchronize()
    if variant is not None:
        H.holder_launch(variant, k, K * 100.0, ctypes.c_void_p(side.cuda_stream))
        time.sleep(0.02)
    t0 = time.perf_counter()
    for _ in range(K): net.forward(x, want_heat=False, decode_size=(540, 960))
    torch.cuda.current_stream().synchronize()
    dt = (time.perf_counter() - t0) / K * 1e3
    torch.cuda.synchronize()
    return dt
print(f'alone: {run():.2f} ms per step')
names = {0: 'light (few registers, sleeping)', 1: 'fat wave (512 registers, sleeping)', 2: 'fat 4-wave workgroup', 3: 'fat wave, fp64 busy loop', 4: 'fat wave, fp64 busy loop + 4.6 KB scratch per lane'}
for variant in (3, 4):
    for k in (1, 8, 16):
        print(f'{names[variant]}, k={k}: {run(variant, k):.2f} ms per step', flush=True)
print(f'alone again: {run():.2f} ms per step')
