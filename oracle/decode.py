"""Oracle: heatmap decodes (numpy).  TEST INFRASTRUCTURE ONLY.

D1  keypoint decode  -- /root/reference/src/models/hrnet/transforms.py:224-239
L2  line 2-peak decode -- /root/reference/src/models/line/transforms.py:224-280

Both are pinned by tests/golden/decode_*.npz captured from the imported reference transforms.

exp definition.  The reference takes ``torch.exp`` (fp32) *before* the max reductions, so distinct
log-probabilities may collide after exp and the first-occurrence rule then picks a different
column / row.  torch's CPU expf (Sleef, 1 ULP) cannot be reproduced bit-for-bit on a GPU, so the
build defines  exp_ref(x) = float32(exp(float64(x)))  (correctly rounded up to double rounding) for
BOTH this oracle and the HIP kernel; the golden vectors confirm the indices agree with the
reference's own exp on the golden inputs, and ``conf`` agrees to 1 ULP.
"""
import numpy as np


def exp_ref(x: np.ndarray) -> np.ndarray:
    return np.exp(x.astype(np.float64)).astype(np.float32)


def keypoint_decode(logp: np.ndarray, img_hw=(540, 960)) -> np.ndarray:
    """transforms.py:228-239.  logp (B,C,h,w) fp32 -> (B,C-1,3) fp32 [x_px, y_px, conf]."""
    B, C, h, w = logp.shape
    H, W = img_hw
    p = exp_ref(logp)
    colmax = p.max(axis=2)                 # (B,C,w)  max over rows      (:231 inner max, dim=2)
    x_prob = colmax.max(axis=2)
    x = colmax.argmax(axis=2)              # first occurrence, like torch.max
    rowmax = p.max(axis=3)                 # (B,C,h)
    y_prob = rowmax.max(axis=2)
    y = rowmax.argmax(axis=2)
    conf = np.minimum(x_prob, y_prob)
    # int64 * python int / python int -> true division in float32 (torch default dtype)
    xf = (x * W).astype(np.float32) / np.float32(w)
    yf = (y * H).astype(np.float32) / np.float32(h)
    out = np.stack([xf, yf, conf.astype(np.float32)], axis=-1)[:, :-1, :]
    return out.astype(np.float32)


def keypoint_decode_indices(logp: np.ndarray):
    """Integer (x, y) argmax indices of keypoint_decode (B,C-1) each."""
    p = exp_ref(logp)
    x = p.max(axis=2).argmax(axis=2)
    y = p.max(axis=3).argmax(axis=2)
    return x[:, :-1], y[:, :-1]


def line_decode(heat: np.ndarray, sigma: float, scale: float) -> np.ndarray:
    """line/transforms.py:193-280.  heat (B,C,H,W) fp32 -> (B,C,2,3) fp32 [x, y, p] * scale on x,y.

    Per (b,c): relu; flat argmax (first occurrence) -> peak 1; multiply by 1-exp(-d^2/(2 sigma^2))
    (all fp32, same operation order as the reference: ((x-x1)^2 + (y-y1)^2) / (2*sigma^2), negate,
    exp, 1-mask, multiply); flat argmax -> peak 2.
    """
    B, C, H, W = heat.shape
    out = -np.ones((B, C, 2, 3), dtype=np.float32)
    t = np.maximum(heat, np.float32(0))
    xs = np.arange(W, dtype=np.float32)[None, :]
    ys = np.arange(H, dtype=np.float32)[:, None]
    two_s2 = np.float32(2.0 * sigma ** 2)
    for b in range(B):
        for c in range(C):
            hm = t[b, c]
            i1 = int(hm.reshape(-1).argmax())
            x1, y1 = np.float32(i1 % W), np.float32(i1 // W)
            out[b, c, 0] = (x1, y1, hm.reshape(-1)[i1])
            d2 = (xs - x1) ** 2 + (ys - y1) ** 2
            mask = exp_ref(-(d2) / two_s2)
            hm2 = hm * (np.float32(1) - mask)
            i2 = int(hm2.reshape(-1).argmax())
            out[b, c, 1] = (np.float32(i2 % W), np.float32(i2 // W), hm2.reshape(-1)[i2])
    out[..., 0] *= np.float32(scale)
    out[..., 1] *= np.float32(scale)
    return out
