"""Synthetic workload generator for the solve stage (benchmarks / smoke runs).

No trained checkpoints ship with the reference and random weights give flat heatmaps, so end-to-end runs
drive the camera solve with keypoints obtained by projecting the pitch template through plausible broadcast
cameras -- the trick the reference itself holds (commented out) at
/root/reference/src/models/hrnet/metamodel.py:69-75.  Rows look like HRNetPredictionTransform output:
[x_px, y_px, conf] on the 2-px decode grid of a 270x480 heatmap.
"""
import numpy as np

from .camera import Camera, pan_tilt_roll_to_orientation
from .pitch import PITCH_ARRAY


def random_camera(rng: np.random.Generator) -> Camera:
    cam = Camera(960, 540)
    pos = np.array([rng.uniform(-35, 35), rng.uniform(50, 95), rng.uniform(-35, -10)])
    target = np.array([rng.uniform(-40, 40), rng.uniform(-20, 20), 0.0])
    d = target - pos
    pan = np.arctan2(d[0], -d[1])
    tilt = np.arctan2(np.hypot(d[0], d[1]), d[2])
    roll = np.deg2rad(rng.normal(0, 1.0))
    f = float(np.exp(rng.uniform(np.log(900), np.log(5000))))
    cam.position = pos
    cam.rotation = np.transpose(pan_tilt_roll_to_orientation(pan, tilt, roll))
    cam.xfocal_length = cam.yfocal_length = np.float64(f)
    cam.calibration = np.array([[f, 0, 480.0], [0, f, 270.0], [0, 0, 1.0]])
    return cam


def keypoints_for_camera(cam: Camera, rng: np.random.Generator, sigma_px=1.0, outlier_frac=0.03, grid=2.0):
    kp = np.zeros((57, 3), dtype=np.float32)
    for i in range(57):
        q = cam.project_point(PITCH_ARRAY[i])
        vis = q[2] != 0 and 0 <= q[0] < 960 and 0 <= q[1] < 540
        if vis:
            p = q[:2] + rng.normal(0, sigma_px, 2)
            if rng.random() < outlier_frac:
                p = np.array([rng.uniform(0, 959), rng.uniform(0, 539)])
            p = np.round(p / grid) * grid
            kp[i] = (min(max(p[0], 0), 960 - grid), min(max(p[1], 0), 540 - grid), rng.uniform(0.55, 1.0))
        else:
            kp[i] = (0.0, 0.0, rng.uniform(0.0, 0.15))
    return kp


def synthetic_keypoints(n: int, seed: int = 0, min_visible: int = 8) -> np.ndarray:
    """(n,57,3) float32 keypoint rows; cameras are re-drawn until at least `min_visible` points show."""
    rng = np.random.Generator(np.random.PCG64(seed))
    out = np.zeros((n, 57, 3), dtype=np.float32)
    for b in range(n):
        while True:
            kp = keypoints_for_camera(random_camera(rng), rng)
            if (kp[:, 2] > 0.5).sum() >= min_visible:
                break
        out[b] = kp
    return out


# ---- peaked-heatmap workload ------------------------------------------------------------------------------------
# Random-init weights give flat, noise-like heatmaps: their argmax says nothing about a trained network and the decoded
# keypoints cannot drive the camera solve.  The construction below keeps the whole random network (so every kernel runs
# on realistic activations and reduced-precision error accumulates through all ~300 layers) and adds ONE designed signal
# path through layers the reference network really has:
#   * each keypoint class k owns a balanced +-1 code over a 2x2x3 input cell; frames carry, around the projected
#     position of keypoint k, cells  0.5 + 0.5 * env * code_k  with a Gaussian envelope (sigma in heatmap cells);
#   * stem filter k (model.conv1, taps of one stride-2 cell only) is the matched filter of code k, bn1 + ReLU cut it at
#     half the matched response, so stem channel k = relu(2 env - 1) near keypoint k and 0 elsewhere (other codes
#     correlate at <= 1/3, and the uniform-noise background (half contrast) stays 6 sigma below the cut);
#   * the head passes stem channel k -> hidden k -> logit k with gain `peak_logit` (the stem features ARE an input of
#     last_layer.0 in the reference: hrnet.py:489-509), every other weight stays random and adds class- and
#     pixel-dependent noise scaled by `noise_gain`; the background class sits at peak_logit / 2.
# Result: heatmaps peaked (p ~ 0.99) on known cells for visible keypoints, background-dominated (p ~ e^-6) for the
# others, riding on the random network's noise -- what HRNetPredictionTransform and CameraCreator see from a trained net.

def stamp_codes() -> np.ndarray:
    """(57,3,2,2) float32 +-1: the first 57 balanced codes of length 12 with pairwise Hamming distance >= 4, in
    lexicographic order of their +1 positions (pairwise correlation <= 1/3)."""
    import itertools
    codes = []
    for pos in itertools.combinations(range(12), 6):
        v = -np.ones(12, dtype=np.int64)
        v[list(pos)] = 1
        if all(int((v != c).sum()) >= 4 for c in codes):
            codes.append(v)
            if len(codes) == 57:
                break
    return np.stack(codes).reshape(57, 3, 2, 2).astype(np.float32)


def peaked_state_dict(sd: dict, peak_logit: float = 12.0, noise_gain: float = 0.25, deep: bool = False, **deep_kw) -> dict:
    """Copy of the random-init HRNet state dict `sd` (keypoint net, 58 classes, stem 64) with the signal path installed.
    deep=False: stem channel k -> hidden k -> logit k (the head reads the stem features directly; the backbone only adds noise).
    deep=True:  the keypoint code travels THROUGH the backbone and reaches the head through the upsampled branch channels only
                (deep_state_dict below); the stem columns of last_layer.0 are zero for the signal rows."""
    import torch
    if deep:
        return deep_state_dict(sd, peak_logit=peak_logit, noise_gain=noise_gain, **deep_kw)
    out = {k: v.clone() for k, v in sd.items()}
    codes = torch.from_numpy(stamp_codes())
    w = out['model.conv1.weight']                       # (64,3,3,3), stride 2, pad 1: taps 1..2 see cell (2i..2i+1, 2j..2j+1)
    w[:57] = 0.0
    w[:57, :, 1:, 1:] = codes
    for key, val in (('weight', 1.0 / 3.0), ('bias', 0.0), ('running_mean', 3.0), ('running_var', 1.0 - 1e-5)):
        out[f'model.bn1.{key}'][:57] = val              # (resp - 3) / 3: matched response 6 -> 1, cut at half
    w0 = out['model.last_layer.0.weight']               # (784,784,1,1); concat order: stem channels first
    w0[:57] = 0.0
    w0[torch.arange(57), torch.arange(57), 0, 0] = 1.0
    out['model.last_layer.0.bias'][:57] = 0.0
    for key, val in (('weight', 1.0), ('bias', 0.0), ('running_mean', 0.0), ('running_var', 1.0 - 1e-5)):
        out[f'model.last_layer.1.{key}'][:57] = val
    w1 = out['model.last_layer.3.weight']               # (58,784,1,1)
    w1 *= noise_gain
    w1[:, :57] = 0.0
    w1[torch.arange(57), torch.arange(57), 0, 0] = peak_logit
    b1 = out['model.last_layer.3.bias']
    b1 *= noise_gain
    b1[57] += peak_logit / 2.0
    return out


# ---- deep signal path ---------------------------------------------------------------------------------------------
# The shallow path above decides a peak's position in the stem conv and the head; 55 % of a step's GPU time (the two-team
# convolutions, the fused 48-channel blocks, the fuse sums, every fp8 layer) only adds noise to it.  The deep path routes
# keypoint class k through channel k of EVERY backbone tensor, on layers the reference network really has, so that the
# position of a heatmap peak is decided by tensors those kernels wrote (hrnet.py:437-511 forward; BasicBlock :42-58;
# fuse layers :183-246; transitions :357-391):
#   conv1 (stem, stride 2)         matched filter of code k, as above                     -> stem k = relu(2 env - 1) at 270x480
#   conv2 (stride 2)               row k: tent kernel on stem channel k, gain `amp`       -> channel k of the 135x240 map
#   layer1.0.downsample            row k: identity on channel k (the Bottleneck chain then carries it on its residual path)
#   transition1.0 / .1             row k: centre tap (k < 48) / stride-2 tent (k < 57)     -> branch 0 (48 ch: classes 0..47) and 1
#   transition2.2, transition3.3   row k: stride-2 tent from the branch above              -> branches 2 and 3 (all 57 classes)
#   every BasicBlock               conv1[k,k] += a1, conv2[k,k] += a2 (BN-compensated centre taps) ON TOP of the random rows:
#                                  out_k = relu(x_k + a2 relu(a1 x_k + noise) + noise): the code passes through the residual
#                                  add AND through both multiplies; (1 + a1 a2)^32 = 4.8 over the 32 blocks of a branch, so
#                                  79 % of what arrives has been through at least one conv multiply
#   every fuse layer               row k: 1x1 [k,k] = fuse_gain (up) / stride-2 tent chain with gain fuse_gain (down): the code
#                                  of class k is exchanged between the resolutions exactly as HRNet exchanges features
#   last_layer.0                   hidden k reads channel k of up(branch 0..3) (concat columns 64+, NOT the stem's), minus a cut
#   last_layer.3                   logit k = gain_k x hidden k; everything else random x noise_gain, as above
# The random rows of every BasicBlock convolution keep feeding pixel- and class-dependent noise into channel k (that is what
# makes near-ties), rows k of the transition / fuse / downsample convolutions are replaced (an exact carry).  A peak now sits
# on the fine-grid cell next to a coarse-grid node (4 px grid for classes 0..47, which ride branch 0; 8 px for 48..56), no
# longer exactly on the stamped cell.  gain_k comes from a noise-free single-channel simulation of the same path.

_TENT = np.array([[0.25, 0.5, 0.25], [0.5, 1.0, 0.5], [0.25, 0.5, 0.25]], dtype=np.float32)       # sum 4: unit DC gain at stride 2 is _TENT / 4


def _signal_peak(has_b0: bool, amp: float, a12: float, fg: float, sigma_cells: float = 2.0) -> float:
    """Noise-free response of ONE signal channel at the head (sum over the four upsampled branches) to a stamp of unit stem
    amplitude, on a 128x128 crop of the stem grid; the stamp sits where the coarse grids line up (cell 64, 64)."""
    import torch
    import torch.nn.functional as F
    n = 128
    yy, xx = np.mgrid[0:n, 0:n]
    env = np.exp(-((yy - 64) ** 2 + (xx - 64) ** 2) / (2.0 * sigma_cells ** 2))
    stem = torch.from_numpy(np.maximum(2.0 * env - 1.0, 0.0).astype(np.float32))[None, None]
    tent = torch.from_numpy(_TENT / 4.0)[None, None]

    def down(t):
        return F.conv2d(t, tent, stride=2, padding=1)

    def up(t, size):
        return F.interpolate(t, size=size, mode='bilinear', align_corners=True)
    s = amp * down(stem)
    b = [s if has_b0 else torch.zeros_like(s), down(s)]
    blocks = (1.0 + a12) ** 4
    for nb, nm in ((2, 1), (3, 4), (4, 3)):
        if len(b) < nb:
            b.append(down(b[-1]))
        for _ in range(nm):
            b = [t * blocks for t in b]
            out = []
            for i in range(nb):
                y = b[i].clone()
                for j in range(nb):
                    if j > i:
                        y = y + fg * up(b[j], b[i].shape[-2:])
                    elif j < i:
                        t = b[j]
                        for q in range(i - j):
                            t = down(t) * (fg if q == i - j - 1 else 1.0)
                        y = y + t
                out.append(y if (i > 0 or has_b0) else torch.zeros_like(y))
            b = out
    h = sum(up(t, (n, n)) for t in b)
    return float(h.max())


def deep_state_dict(sd: dict, peak_logit: float = 12.0, noise_gain: float = 0.25, amp: float = 48.0, a1: float = 0.5,
                    a2: float = -0.04, fuse_gain: float = 0.1, cut: float = 0.25, boost: float = 3.0, row_gain: float = 0.1) -> dict:
    """See the block comment above.  `sd`: random-init HRNet-W48-shaped keypoint state dict (58 classes, stem 64, branch widths
    48 / 96 / 192 / 384 or any widths with >= 57 channels from branch 1 on)."""
    import torch
    out = peaked_state_dict(sd, peak_logit=peak_logit, noise_gain=noise_gain)          # stem filters + last_layer.3 noise scaling
    eps = 1e-5
    tent = torch.from_numpy(_TENT)

    def ident_bn(bn, n):
        for key, val in (('weight', 1.0), ('bias', 0.0), ('running_mean', 0.0), ('running_var', 1.0 - eps)):
            out[f'{bn}.{key}'][:n] = val

    def carry(conv, bn, n, kernel, gain=1.0):
        """rows < n of `conv` := `kernel` (3x3 tensor, or None for a 1x1 / centre tap) x gain on the diagonal; BN identity there."""
        w = out[conv + '.weight']
        n = min(n, w.shape[0])
        w[:n] = 0.0
        m = min(n, w.shape[1])
        idx = torch.arange(m)
        if kernel is None:
            w[idx, idx, w.shape[2] // 2, w.shape[3] // 2] = gain
        else:
            w[idx, idx] = kernel * gain
        ident_bn(bn, n)

    carry('model.conv2', 'model.bn2', 57, tent / 4.0, amp)
    carry('model.layer1.0.downsample.0', 'model.layer1.0.downsample.1', 57, None)
    widths = {}
    for key, v in out.items():           # branch widths from the block convolutions themselves
        if key.startswith('model.stage4.0.branches.') and key.endswith('.0.conv1.weight'):
            widths[int(key.split('.')[4])] = v.shape[0]
    carry('model.transition1.0.0', 'model.transition1.0.1', min(57, widths[0]), None)
    carry('model.transition1.1.0.0', 'model.transition1.1.0.1', 57, tent / 4.0)
    if widths[0] < 57:                  # classes that find no channel in branch 0 enter branch 1 `boost` times stronger
        out['model.transition1.1.0.0.weight'][widths[0]:57] *= boost
    carry('model.transition2.2.0.0', 'model.transition2.2.0.1', 57, tent / 4.0)
    carry('model.transition3.3.0.0', 'model.transition3.3.0.1', 57, tent / 4.0)
    for key in [k for k in out if '.branches.' in k and k.endswith(('.conv1.weight', '.conv2.weight'))]:
        conv = key[:-len('.weight')]
        bn = conv.replace('.conv1', '.bn1').replace('.conv2', '.bn2')
        w = out[key]
        n = min(57, w.shape[0])
        idx = torch.arange(n)
        scale = out[bn + '.weight'][:n] / torch.sqrt(out[bn + '.running_var'][:n] + eps)          # the BN that follows multiplies by this
        if conv.endswith('conv2'):
            w[:n] *= row_gain                       # how strongly the random network (noise + the other classes' codes) writes into channel k
        w[idx, idx, 1, 1] += (a1 if conv.endswith('conv1') else a2) / scale
    for key in [k for k in out if '.fuse_layers.' in k and k.endswith('.0.weight')]:
        conv = key[:-len('.weight')]
        bn = conv[:-1] + '1'
        parts = conv.split('.')                   # model.stageS.M.fuse_layers.I.J.0   or   ...I.J.Q.0
        i, j = int(parts[4]), int(parts[5])
        if j > i:
            carry(conv, bn, 57, None, fuse_gain)
        else:
            q = int(parts[6])
            carry(conv, bn, 57, tent / 4.0, fuse_gain if q == i - j - 1 else 1.0)
    # head: hidden k <- channel k of the upsampled branches (concat order: stem 64 | b0 | b1 | b2 | b3), cut, logit gain
    w0 = out['model.last_layer.0.weight']
    w0[:57] = 0.0
    off = 64
    for b in sorted(widths):
        n = min(57, widths[b])
        idx = torch.arange(n)
        w0[idx, off + idx, 0, 0] = 1.0
        off += widths[b]
    ident_bn('model.last_layer.1', 57)
    w1 = out['model.last_layer.3.weight']
    peaks = {True: _signal_peak(True, amp, a1 * a2, fuse_gain), False: boost * _signal_peak(False, amp, a1 * a2, fuse_gain)}
    for k in range(57):
        pk = peaks[k < widths[0]]
        out['model.last_layer.0.bias'][k] = -cut * pk
        w1[k, k, 0, 0] = peak_logit / ((1.0 - cut) * pk)
    return out


def line_keypoints():
    """{line class index (LINE_CLS order): [keypoint ids that lie on it]} from the 30 line-pair intersections the reference joins
    (LINE_INTERSECTIONS, prediction.py:105-124)."""
    from .lines import LINE_CLS, LINE_INTERSECTIONS
    index = {name: i for i, name in LINE_CLS.items()}
    on = {i: [] for i in LINE_CLS}
    for kp, pair in LINE_INTERSECTIONS.items():
        for name in pair:
            on[index[name]].append(int(kp))
    return on


def line_deep_state_dict(sd_line: dict, peak_logit: float = 12.0, noise_gain: float = 0.25, amp: float = 48.0, a1: float = 0.5,
                         a2: float = -0.04, fuse_gain: float = 0.1, cut: float = 0.25, boost: float = 3.0, row_gain: float = 0.1) -> dict:
    """The C4 workload's LINE network (BASELINE config C4; src/models/line/hrnet.py: the W48 backbone, no stem concat, Softmax over the 23
    line classes at stride 4): random-init `sd_line` with the SAME deep signal path as the keypoint network's (deep_state_dict: keypoint
    class k rides channel k of every backbone tensor -- the frames carry the keypoint stamps, both networks see the same frames) and a
    head that answers with LINES: hidden k reads channel k of the upsampled branches, the logit of line l sums the hidden units of the
    keypoints that lie on l (line_keypoints).  At a visible keypoint the two lines that cross there get ~peak_logit each (softmax 0.5 /
    0.5), elsewhere all 23 logits are the random network's noise (~1/23 each): the two-peak decode (EHMPredictionTransform) finds two
    keypoints of every line that shows two, get_line_data (prob_thre 0.2, its own default) drops the others, and the line-pair
    intersections land on the keypoints the keypoint network decodes -- line points CONSISTENT with the frame, as a trained line network
    gives.  (The round 2-5 C4 bench used the raw random line network: 28 garbage line points per frame, 21 of them inside the image, which
    original_voter adds to every frame's point set (prediction.py:356-364); two frames in three then failed the first pass and every
    candidate fit crawled to 20000 iterations -- a solve stage of 320 ms per batch on the masked CUs, profiles/README.md round 6.)"""
    import torch
    tmp = {k: v.clone() for k, v in sd_line.items()}
    # deep_state_dict installs the backbone path and writes a keypoint-shaped head: give it scratch head tensors, keep its backbone
    for key, shape in (('model.last_layer.0.weight', (784, 784, 1, 1)), ('model.last_layer.0.bias', (784,)), ('model.last_layer.1.weight', (784,)),
                       ('model.last_layer.1.bias', (784,)), ('model.last_layer.1.running_mean', (784,)), ('model.last_layer.1.running_var', (784,)),
                       ('model.last_layer.3.weight', (58, 784, 1, 1)), ('model.last_layer.3.bias', (58,))):
        tmp[key] = torch.ones(shape) if key.endswith('running_var') else torch.zeros(shape)
    deep = deep_state_dict(tmp, peak_logit=peak_logit, noise_gain=noise_gain, amp=amp, a1=a1, a2=a2, fuse_gain=fuse_gain, cut=cut, boost=boost,
                           row_gain=row_gain)
    out = {k: (sd_line[k].clone() if k.startswith('model.last_layer.') else deep[k]) for k in sd_line}
    widths = {}
    for key, v in out.items():
        if key.startswith('model.stage4.0.branches.') and key.endswith('.0.conv1.weight'):
            widths[int(key.split('.')[4])] = v.shape[0]
    eps = 1e-5
    w0 = out['model.last_layer.0.weight']                  # (720,720,1,1); concat order b0 | b1 | b2 | b3 (no stem: upscale 1)
    w0[:57] = 0.0
    off = 0
    for b in sorted(widths):
        n = min(57, widths[b])
        idx = torch.arange(n)
        w0[idx, off + idx, 0, 0] = 1.0
        off += widths[b]
    for key, val in (('weight', 1.0), ('bias', 0.0), ('running_mean', 0.0), ('running_var', 1.0 - eps)):
        out[f'model.last_layer.1.{key}'][:57] = val
    peaks = {True: _signal_peak(True, amp, a1 * a2, fuse_gain), False: boost * _signal_peak(False, amp, a1 * a2, fuse_gain)}
    w1 = out['model.last_layer.3.weight']                  # (23,720,1,1)
    w1 *= noise_gain
    w1[:, :57] = 0.0
    out['model.last_layer.3.bias'] *= noise_gain
    on = line_keypoints()
    for k in range(57):
        pk = peaks[k < widths[0]]
        out['model.last_layer.0.bias'][k] = -cut * pk
        for line, kps in on.items():
            if k in kps:
                w1[line, k, 0, 0] = peak_logit / ((1.0 - cut) * pk)
    return out


def stamped_frames(n: int, seed: int = 0, sigma_cells: float = 2.0, jitter_px: float = 1.0, min_visible: int = 8,
                   size=(540, 960)):
    """(frames (n,3,H,W) float32 in [0,1] -- uniform-noise background in [0.25, 0.75) with the keypoint stamps --, expect (n,57,3) float32
    rows [x_px, y_px, visible] in 960x540 units on the decode grid of an (H/2, W/2) heatmap -- 2 px at 960x540, 1 px at
    1920x1080: HRNetPredictionTransform scales by size=[540,960] whatever the input size, transforms.py:234-235).
    Cameras are re-drawn until `min_visible` keypoints show."""
    H, W = size
    rng = np.random.Generator(np.random.PCG64(seed))
    codes = stamp_codes()                                # (57,3,2,2)
    frames = 0.25 + 0.5 * rng.random((n, 3, H, W), dtype=np.float32)      # matched-filter response of the background: sigma 0.5, the cut sits at 6 sigma
    expect = np.zeros((n, 57, 3), dtype=np.float32)
    hc, wc = H // 2, W // 2
    R = int(np.ceil(2.4 * sigma_cells))                  # env < 1/2 beyond 1.18 sigma; a little margin
    dy, dx = np.mgrid[-R:R + 1, -R:R + 1]
    env0 = np.exp(-(dy * dy + dx * dx) / (2.0 * sigma_cells ** 2)).astype(np.float32)
    for b in range(n):
        while True:
            cam = random_camera(rng)
            q = cam.project_points(PITCH_ARRAY)
            vis = (q[:, 2] != 0) & (q[:, 0] >= 0) & (q[:, 0] < 960) & (q[:, 1] >= 0) & (q[:, 1] < 540)
            if vis.sum() >= min_visible:
                break
        owner_env = np.zeros((hc, wc), dtype=np.float32)
        cells = frames[b].reshape(3, hc, 2, wc, 2)       # view: [c, i, ky, j, kx]
        for k in np.nonzero(vis)[0]:
            p = q[k, :2] + rng.normal(0, jitter_px, 2)
            cj = int(min(max(round(p[0] * wc / 960.0), 0), wc - 1))         # cell of the (hc, wc) stem / heatmap grid
            ci = int(min(max(round(p[1] * hc / 540.0), 0), hc - 1))
            i0, i1, j0, j1 = max(ci - R, 0), min(ci + R, hc - 1), max(cj - R, 0), min(cj + R, wc - 1)
            env = env0[i0 - ci + R:i1 - ci + R + 1, j0 - cj + R:j1 - cj + R + 1]
            take = env > owner_env[i0:i1 + 1, j0:j1 + 1]                  # overlapping blobs: the stronger envelope owns the cell
            owner_env[i0:i1 + 1, j0:j1 + 1] = np.where(take, env, owner_env[i0:i1 + 1, j0:j1 + 1])
            stamp = 0.5 + 0.5 * env[None, :, None, :, None] * codes[k][:, None, :, None, :]     # (3,ni,2,nj,2)
            blk = cells[:, i0:i1 + 1, :, j0:j1 + 1, :]
            blk[...] = np.where(take[None, :, None, :, None], stamp, blk)
            expect[b, k] = (cj * 960.0 / wc, ci * 540.0 / hc, 1.0)
    return frames, expect


# ---- synthetic annotations (N4 geometry tests): the pitch's lines and circles as a SoccerNet annotator would click them ----
def rescaled_state_dict(sd: dict, conv_units, seed: int = 0, sigma_log2: float = 3.0, dead_frac: float = 0.01, dead_log2: int = -12,
                         max_log2: int = 10, only=None, k_fixed=None) -> dict:
    """A "trained-like" spread of scales WITHOUT changing the function: for every block-internal BatchNorm (bn1 of a BasicBlock, bn1 and bn2
    of a Bottleneck -- its output feeds exactly one convolution) channel c gets s_c = 2^k_c: gamma_c, beta_c *= s_c and column c of the
    next convolution's weight /= s_c.  ReLU commutes with a positive scale and powers of two are exact, so the fp32 network's output is
    BIT-identical (hrnet.py:42-58, 79-99), while the internal activations and the folded weights now span decades the way a trained
    checkpoint's do: k_c ~ round(N(0, sigma_log2)) clipped to [-max_log2, max_log2] (sigma 3 -> about four decades), and a fraction
    `dead_frac` of near-dead channels at 2^dead_log2.  `conv_units` = HRNetHeatmap.conv_units(); `only` = substring filter on the
    first convolution's name; `k_fixed` = the same exponent for every channel (the scale-equivariance test).  Workload / test generator."""
    import torch
    rng = np.random.Generator(np.random.PCG64(seed))
    out = {k: v.clone() for k, v in sd.items()}
    units = list(conv_units)
    for i in range(len(units) - 1):
        name, bn, cin, cout = units[i][0], units[i][1], units[i][2], units[i][3]
        nxt = units[i + 1][0]
        stem, leaf = name.rsplit('.', 1)
        nstem, nleaf = nxt.rsplit('.', 1)
        if not bn or stem != nstem or stem == 'model' or (leaf, nleaf) not in (('conv1', 'conv2'), ('conv2', 'conv3')):      # (the stem's bn1 also feeds the head)
            continue
        if only is not None and only not in name:
            continue
        if k_fixed is not None:
            k = np.full(cout, int(k_fixed))
        else:
            k = np.clip(np.rint(rng.normal(0.0, sigma_log2, cout)), -max_log2, max_log2).astype(np.int64)
            k[rng.random(cout) < dead_frac] = dead_log2
        s_c = torch.from_numpy(np.exp2(k.astype(np.float64))).to(out[bn + '.weight'].dtype)
        out[bn + '.weight'] = out[bn + '.weight'] * s_c
        out[bn + '.bias'] = out[bn + '.bias'] * s_c
        w = out[nxt + '.weight']
        out[nxt + '.weight'] = w / s_c.to(w.dtype).view(1, -1, 1, 1)
    return out


def synthetic_annotation(seed: int = 0, pts_per_line: int = 6, noise_px: float = 0.3):
    """{annotation class: [(x, y) normalised to the 960x540 image]} for one sampled camera: every pitch line that runs through
    at least two template intersections (sampled between its extreme intersection points) and the three circles, projected,
    clipped to the image, with click noise.  Also returns the camera."""
    from .lines import LINE_INTERSECTIONS
    from .pitch import INTERSECTON_TO_PITCH_POINTS, PITCH_POINTS
    rng = np.random.Generator(np.random.PCG64(seed))
    cam = random_camera(rng)
    by_line = {}
    for idx, pair in LINE_INTERSECTIONS.items():
        for name in pair:
            by_line.setdefault(name, []).append(np.asarray(PITCH_POINTS[INTERSECTON_TO_PITCH_POINTS[idx]], dtype=np.float64))
    world = {}
    for name, pts in by_line.items():
        pts = np.array(pts)
        if len(pts) < 2:
            continue
        span = pts.max(axis=0) - pts.min(axis=0)
        ax = int(np.argmax(span))
        a, b = pts[np.argmin(pts[:, ax])], pts[np.argmax(pts[:, ax])]
        world[name] = a[None] + np.linspace(0.0, 1.0, 4 * pts_per_line)[:, None] * (b - a)[None]
    R = 9.15
    ang = np.linspace(0, 2 * np.pi, 48, endpoint=False)
    world['Circle central'] = np.c_[R * np.cos(ang), R * np.sin(ang), np.zeros_like(ang)]
    for name, cx, sgn in (('Circle left', -41.5, 1.0), ('Circle right', 41.5, -1.0)):
        arc = np.c_[cx + R * np.cos(ang), R * np.sin(ang), np.zeros_like(ang)]
        world[name] = arc[sgn * (arc[:, 0] - sgn * -36.0) > 0] if sgn > 0 else arc[arc[:, 0] < 36.0]
    out = {}
    for name, w in world.items():
        q = cam.project_points(w)
        ok = (q[:, 2] != 0) & (q[:, 0] >= 0) & (q[:, 0] <= 960) & (q[:, 1] >= 0) & (q[:, 1] <= 540)
        p = q[ok, :2]
        if len(p) < 2:
            continue
        if len(p) > pts_per_line and not name.startswith('Circle'):
            p = p[np.linspace(0, len(p) - 1, pts_per_line).round().astype(int)]
        p = p + rng.normal(0, noise_px, p.shape)
        out[name] = [(float(x) / 960.0, float(y) / 540.0) for x, y in p]
    return out, cam
