"""JPEG input stage timing on the GPU box: host entropy decode per thread count, device kernels,
and the CPU reference point (Pillow = libjpeg-turbo full decode, what cv2.imread costs per frame).
    python tools/jpeg_bench.py [B] [threads ...]
"""
import io
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sncal_amd  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
threads = [int(a) for a in sys.argv[2:]] or [1, 4, 8, 16, 32]
g = np.load(os.path.join(os.path.dirname(__file__), '..', 'tests', 'golden', 'jpeg_cases.npz'))
blob = g['jpg.full'].tobytes()
print('frame: 960x540 4:2:0,', len(blob), 'bytes')
dev = torch.device('cuda:0')
try:
    from PIL import Image
    t0 = time.perf_counter()
    for _ in range(20):
        np.asarray(Image.open(io.BytesIO(blob)).convert('RGB'))
    print('libjpeg-turbo (Pillow) full CPU decode: %.2f ms/frame/core' % ((time.perf_counter() - t0) / 20 * 1e3))
except ImportError:
    pass
t0 = time.perf_counter()
for _ in range(20):
    sncal_amd.jpeg.entropy_decode(blob)
print('host entropy decode alone: %.2f ms/frame/core' % ((time.perf_counter() - t0) / 20 * 1e3))
for nt in threads:
    dec = sncal_amd.JpegDecoder(540, 960, max_batch=B, threads=nt, device=dev)
    out = torch.empty((B, 540, 960, 3), dtype=torch.uint8, device=dev)
    blobs = [blob] * B
    for _ in range(2):
        dec.decode(blobs, out)
    torch.cuda.synchronize()
    n = 5
    t0 = time.perf_counter()
    for _ in range(n):
        dec.decode(blobs, out)
    t_host = (time.perf_counter() - t0) / n
    torch.cuda.synchronize()
    t_all = (time.perf_counter() - t0) / n
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    print('threads %2d: host part %.2f ms/batch of %d (%.0f frames/s), with device drain %.2f ms (%.0f frames/s)'
          % (nt, t_host * 1e3, B, B / t_host, t_all * 1e3, B / t_all))
    dec.close()
