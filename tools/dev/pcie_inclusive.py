"""PCIe-inclusive rate: frames start in PINNED HOST memory as cv2-style uint8 (B,540,960,3); every step
uploads its batch on a copy stream (double-buffered) and runs the whole pipeline (forward_u8 + decode + solves).
Also the JPEG variant: encoded frames in host memory -> JpegDecoder (host Huffman threads + device kernels) -> pipeline."""
import os, sys, time
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import sncal_amd
from bench import seeded_weights
dev = torch.device('cuda:0')
B, K = 64, 8
net = sncal_amd.HRNetHeatmap('hrnet_w48', dtype='bf16', device=dev)
net.load_state_dict(seeded_weights('hrnet_w48', 1))
cc = sncal_amd.submit.default_calibrator()
pipe = sncal_amd.CalibrationPipeline(net, cc, decode_size=(540, 960))
kp = torch.from_numpy(sncal_amd.synth.synthetic_keypoints(B, seed=77)).to(dev)
host = [torch.randint(0, 256, (B, 540, 960, 3), dtype=torch.uint8).pin_memory() for _ in range(2)]
devb = [torch.empty((B, 540, 960, 3), dtype=torch.uint8, device=dev) for _ in range(2)]
copy = torch.cuda.Stream(device=dev)
main = torch.cuda.current_stream(dev)
ev_up = [torch.cuda.Event() for _ in range(2)]
ev_done = [torch.cuda.Event() for _ in range(2)]


def run(steps):
    for k in range(steps):
        s = k & 1
        with torch.cuda.stream(copy):
            copy.wait_event(ev_done[s])                 # the forward that last read this device buffer is over
            devb[s].copy_(host[s], non_blocking=True)
            ev_up[s].record(copy)
        main.wait_event(ev_up[s])
        pipe.submit(devb[s], extra_keypoints=kp)
        ev_done[s].record(main)
    pipe.join(); torch.cuda.synchronize()


run(3)
t0 = time.perf_counter(); run(K); dt = (time.perf_counter() - t0) / K
print(f'uint8 frames from pinned host memory: {dt*1e3:.1f} ms/step, {B/dt:.0f} frames/s (H2D {B*540*960*3/1e6:.0f} MB per step)')

g = np.load(os.path.join(ROOT, 'tests', 'golden', 'jpeg_cases.npz'))
blob = g['jpg.full'].tobytes()
dec = sncal_amd.JpegDecoder(540, 960, max_batch=B, threads=16, device=dev)


def run_jpeg(steps):
    for k in range(steps):
        x = dec.decode([blob] * B, devb[k & 1])
        pipe.submit(x, extra_keypoints=kp)
    pipe.join(); torch.cuda.synchronize()


run_jpeg(3)
t0 = time.perf_counter(); run_jpeg(K); dt = (time.perf_counter() - t0) / K
print(f'JPEG bytes in host memory (16 Huffman threads, in line with the submit loop): {dt*1e3:.1f} ms/step, {B/dt:.0f} frames/s')
