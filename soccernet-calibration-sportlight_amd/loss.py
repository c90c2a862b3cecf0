"""Training-target synthesis on the GPU (heatmap half): create_heatmaps / HRNetLoss.create_target of
/root/reference/src/models/hrnet/loss.py:21-52, 81-87, same names and arguments.  The losses themselves (training)
are out of scope."""
from typing import Tuple

import torch

from . import _lib


def create_target(keypoints: torch.Tensor, sigma: float, pred_size: Tuple[int, int] = (68, 120)) -> torch.Tensor:
    """HRNetLoss.create_target (loss.py:81-87): keypoints (B,N,3) [x, y, visibility] in heatmap pixels ->
    (B,N+1,H,W) fp32 heatmaps, last channel = 1 - max over the keypoint channels."""
    kp = _lib.require_device(keypoints.contiguous(), torch.float32, 'keypoints')
    if kp.dim() != 3 or kp.shape[2] != 3:
        raise _lib.SncalError('keypoints must be (B,N,3)')
    B, N = kp.shape[0], kp.shape[1]
    h, w = int(pred_size[0]), int(pred_size[1])
    out = torch.empty((B, N + 1, h, w), dtype=torch.float32, device=kp.device)
    with torch.cuda.device(kp.device):
        _lib.check(_lib.lib().sncal_create_target(kp.data_ptr(), B, N, float(sigma), h, w, out.data_ptr(),
                                                  _lib.current_stream_ptr()), 'sncal_create_target')
    return out


def create_heatmaps(keypoints: torch.Tensor, sigma: float, pred_size: Tuple[int, int] = (68, 120)) -> torch.Tensor:
    """loss.py:21-52: (B,N,H,W) Gaussian heatmaps.  keypoints (B,N,2) or (B,N,3); visibility is the reference's test
    any(keypoints == 1, dim=-1) over the components given."""
    kp = keypoints
    if kp.shape[-1] == 2:          # no third component: it cannot make a point visible
        kp = torch.cat([kp, torch.zeros_like(kp[..., :1])], dim=-1)
    return create_target(kp, sigma, pred_size)[:, :-1]
