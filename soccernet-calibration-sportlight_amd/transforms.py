"""Prediction transforms (heatmap decodes) -- host mirror of the reference classes, running on HIP.

HRNetPredictionTransform  <->  /root/reference/src/models/hrnet/transforms.py:224-239
EHMPredictionTransform    <->  /root/reference/src/models/line/transforms.py:193-280
Same constructor arguments, same call signature, same output layout; the arithmetic runs in
libsncal.so (csrc/decode.hip).
"""
import torch

from . import _lib


class HRNetPredictionTransform:
    def __init__(self, size):
        self.H, self.W = size

    def __call__(self, preds: torch.Tensor) -> torch.Tensor:
        """preds (B,58,h,w) fp32 log-probabilities on the GPU -> (B,57,3) fp32 [x_px, y_px, conf]."""
        _lib.require_device(preds, torch.float32, 'preds')
        B, C, h, w = preds.shape
        out = torch.empty((B, C - 1, 3), dtype=torch.float32, device=preds.device)
        with torch.cuda.device(preds.device):
            _lib.check(_lib.lib().sncal_heatmap_decode(preds.data_ptr(), B, C, h, w, int(self.H), int(self.W),
                                                       out.data_ptr(), _lib.current_stream_ptr()),
                       'sncal_heatmap_decode')
        return out


class EHMPredictionTransform:
    def __init__(self, scale=8, sigma=6):
        self.scale = scale
        self.sigma = sigma
        self.distance_threshold = 2 * self.sigma

    def __call__(self, preds: torch.Tensor) -> torch.Tensor:
        """preds (B,N,H,W) fp32 line heatmaps on the GPU -> (B,N,2,3) fp32 [x*scale, y*scale, p]."""
        return self._decode(preds, self.sigma, self.scale)

    @staticmethod
    def mask_heat_points_gauss(tensor: torch.Tensor, sigma: float = 5) -> torch.Tensor:
        return EHMPredictionTransform._decode(tensor, sigma, 1.0)

    @staticmethod
    def _decode(t, sigma, scale):
        _lib.require_device(t, torch.float32, 'preds')
        B, C, H, W = t.shape
        out = torch.empty((B, C, 2, 3), dtype=torch.float32, device=t.device)
        with torch.cuda.device(t.device):
            _lib.check(_lib.lib().sncal_line_decode(t.data_ptr(), B, C, H, W, float(sigma), float(scale),
                                                    out.data_ptr(), _lib.current_stream_ptr()),
                       'sncal_line_decode')
        return out
