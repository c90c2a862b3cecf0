"""JPEG input stage: the `cv2.imread` of /root/reference/src/utils/make_submit.py:62 and
/root/reference/src/utils/export_line_result.py:176, producing the BGR uint8 frames on the GPU.

Host threads do what is bit-serial (markers, Huffman); the device does the inverse DCT, chroma upsampling and
colour conversion exactly as libjpeg-turbo's defaults (what cv2 wraps) compute them.  See csrc/jpeg.hip.
"""
import ctypes
from typing import Sequence

import numpy as np
import torch

from . import _lib


def probe(data: bytes) -> dict:
    """Header fields of one JPEG (host only)."""
    info = _lib.JpegInfo()
    _lib.check(_lib.lib().sncal_jpeg_probe(data, len(data), ctypes.byref(info)), 'sncal_jpeg_probe')
    return dict(width=info.width, height=info.height, components=info.components, h_samp=info.h_samp,
                v_samp=info.v_samp, restart_interval=info.restart_interval, blocks=list(info.blocks))


def entropy_decode(data: bytes):
    """Quantised coefficient blocks of one JPEG, per component (block_rows, block_cols, 64) int16 in natural
    order (host only; the layout the device kernels consume)."""
    info = probe(data)
    n = sum(info['blocks']) * 64
    buf = np.zeros(n, np.int16)
    ci = _lib.JpegInfo()
    _lib.check(_lib.lib().sncal_jpeg_entropy_decode(data, len(data), buf.ctypes.data, n, ctypes.byref(ci)),
               'sncal_jpeg_entropy_decode')
    hs, vs = info['h_samp'], info['v_samp']
    mx = -(-info['width'] // (8 * hs))
    my = -(-info['height'] // (8 * vs))
    out, off = [], 0
    for c in range(info['components']):
        bh, bw = (my * vs, mx * hs) if c == 0 else (my, mx)
        out.append(buf[off:off + bh * bw * 64].reshape(bh, bw, 64))
        off += bh * bw * 64
    return out


class JpegDecoder:
    """Batched decoder for frames of one size.  decode(list of bytes) -> (B,H,W,3) uint8 BGR CUDA tensor, the
    stacked `cv2.imread` results; feed it to HRNet.forward / CalibrationPipeline.submit directly.  Calls on one
    decoder must use one stream (its device buffers are reused in stream order)."""

    def __init__(self, height: int, width: int, max_batch: int = 64, threads: int = 0, device='cuda:0'):
        self.device = torch.device(device)
        self.height, self.width, self.max_batch = int(height), int(width), int(max_batch)
        h = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().sncal_jpeg_create(self.max_batch, self.height, self.width, int(threads),
                                                    ctypes.byref(h)), 'sncal_jpeg_create')
        self._h = h

    def decode(self, frames: Sequence[bytes], out: torch.Tensor = None) -> torch.Tensor:
        B = len(frames)
        if out is None:
            out = torch.empty((B, self.height, self.width, 3), dtype=torch.uint8, device=self.device)
        _lib.require_device(out, torch.uint8, 'out')
        if tuple(out.shape) != (B, self.height, self.width, 3):
            raise _lib.SncalError(f'out must be ({B},{self.height},{self.width},3)')
        ptrs = (ctypes.c_char_p * max(B, 1))(*[bytes(f) for f in frames])
        lens = (ctypes.c_size_t * max(B, 1))(*[len(f) for f in frames])
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().sncal_jpeg_decode(self._h, ptrs, lens, B, out.data_ptr(), _lib.current_stream_ptr()),
                       'sncal_jpeg_decode')
        return out

    def decode_files(self, paths: Sequence[str], out: torch.Tensor = None) -> torch.Tensor:
        blobs = []
        for p in paths:
            with open(p, 'rb') as f:
                blobs.append(f.read())
        return self.decode(blobs, out)

    def close(self):
        if self._h:
            _lib.lib().sncal_jpeg_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
