"""HRNet keypoint / line networks on libsncal.so -- host mirror of the reference modules.

HRNetHeatmap  <->  /root/reference/src/models/hrnet/model.py:130-150 and
                   /root/reference/src/models/line/model.py:127-147 (num_refinement_stages == 0, the only
                   value any shipped config uses: train_config.yaml:34, val_config.yaml:27)
The object is callable like the reference nn.Module (``net(x)[-1]`` is the (B,C,h,w) head output) and loads
the reference's state dict (``model.`` prefix, SyncBatchNorm running stats) -- eval-mode BatchNorm is folded
into the conv weights before they are packed for the MFMA kernels.  PyTorch only provides device memory.
"""
import ctypes
import os
from collections.abc import Mapping, Sequence

import numpy as np
import torch
import yaml

from . import _lib

_CFG_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'configs')
BN_EPS = 1e-5
# 'fp16x3' / 'bf16x3': the split-arithmetic (fp32-class) engine; a given libsncal.so implements ONE of the two 16-bit split types
# (sncal_x3_name(): fp16 since round 4, bf16 with -DSNCAL_X3_F16=0) and refuses the other name; 'x3' takes whichever the build has
DTYPES = {'fp32': 0, 'f32': 0, 'float32': 0, 'bf16': 1, 'bfloat16': 1, 'fp8': 2, 'e4m3': 2, 'fp16x3': 3, 'bf16x3': 3, 'x3': 3}


def _plain(obj):
    """Nested mappings / sequences of any kind (dict, OmegaConf DictConfig / ListConfig -- what hydra.utils.instantiate
    leaves in an argus checkpoint's params -- or a yaml tree) -> plain dicts and lists."""
    try:
        from omegaconf import OmegaConf
        if OmegaConf.is_config(obj):
            return OmegaConf.to_container(obj, resolve=True)
    except ImportError:
        pass
    if isinstance(obj, Mapping):
        return {str(k): _plain(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)) or (isinstance(obj, Sequence) and not isinstance(obj, (str, bytes))):
        return [_plain(v) for v in obj]
    return obj


def load_config(name_or_path, head=None, upscale=None):
    """yaml with the reference's field names (src/models/hrnet/model_config/*.yaml), a mapping holding them (the
    reference reads them by attribute from an OmegaConf node, hrnet.py:256-314) or the name of a packaged config.
    `head` / `upscale`: values the CALLING model class fixes in code rather than in the yaml -- the line network
    hard-codes Softmax and has no upscale branch (src/models/line/hrnet.py:86-102, 236-245), its yaml carries neither key."""
    if isinstance(name_or_path, Mapping) or type(name_or_path).__name__ in ('DictConfig',):
        cfg = _plain(name_or_path)
    elif isinstance(name_or_path, (str, os.PathLike)):
        path = name_or_path if os.path.exists(str(name_or_path)) else os.path.join(_CFG_DIR, f'{name_or_path}.yaml')
        if not os.path.exists(path):
            raise _lib.SncalError(f'hrnet_config: no such file or packaged config: {name_or_path!r}')
        with open(path) as f:
            cfg = yaml.safe_load(f)
    else:
        raise _lib.SncalError(f'hrnet_config must be a mapping, a yaml path or a packaged config name, not {type(name_or_path).__name__}')
    if head is not None:
        cfg['head'] = head
    if upscale is not None:
        cfg['upscale'] = upscale
    cfg.setdefault('upscale', 1)
    cfg.setdefault('head', 'logsoftmax')
    return cfg


def _desc(cfg):
    d = _lib.HRNetDesc()
    d.num_classes = int(cfg['num_classes'])
    d.stem_width = int(cfg['stem_width'])
    d.upscale = int(cfg.get('upscale', 1) or 1)
    d.head_softmax = 1 if cfg.get('head', 'logsoftmax') == 'softmax' else 0
    if cfg.get('internal_final_conv', 0):
        raise _lib.SncalError('internal_final_conv != 0 is not supported (unused by every shipped config)')
    if int(cfg.get('final_conv_kernel', 1)) != 1:
        raise _lib.SncalError('final_conv_kernel must be 1')
    s1 = cfg['stage1']
    if s1['block_type'] != 'BOTTLENECK':
        raise _lib.SncalError('stage1 must use BOTTLENECK blocks')
    d.stage1_blocks = int(s1['num_blocks'][0])
    d.stage1_channels = int(s1['num_channels'][0])
    for i, key in enumerate(('stage2', 'stage3', 'stage4')):
        sc = cfg[key]
        if sc['block_type'] != 'BASIC' or len(set(sc['num_blocks'])) != 1:
            raise _lib.SncalError(f'{key}: BASIC blocks with a uniform block count are required')
        d.num_modules[i] = int(sc['num_modules'])
        d.num_branches[i] = int(sc['num_branches'])
        d.num_blocks[i] = int(sc['num_blocks'][0])
        for b, c in enumerate(sc['num_channels']):
            d.num_channels[i][b] = int(c)
    return d


class HRNetHeatmap:
    """Inference-only HRNet on the HIP engine.  ``dtype``:
      None / 'fp16x3'  (default) the fp32-class engine: fp32 tensors and accumulation, every product as hi.hi + hi.lo + lo.hi of fp16
                       splits on the 16-bit matrix pipe -- the engine bench.py measures; on 2048 deep-path frames it returned the exact
                       engine's keypoint indices on 39642 of 39642 usable rows and its camera on 2045 of 2045 frames
                       (profiles/r04_parity_large_*); 'bf16x3' names the same engine built on bf16 splits (-DSNCAL_X3_F16=0)
      'fp32'           the reference's own arithmetic on the exact-fp32 MFMA (v_mfma_f32_16x16x4_f32): 2.8x slower
      'bf16' / 'fp8'   OPT-IN throughput modes (bf16 tensors; e4m3 wide convolutions, BASELINE config C5): 1-3 % of the usable
                       keypoints move by one heatmap cell."""

    def __init__(self, hrnet_config, num_refinement_stages: int = 0, num_heatmaps: int = None,
                 dtype: str = None, device='cuda:0', head=None, upscale=None):
        if num_refinement_stages != 0:
            raise _lib.SncalError('refinement stages are never instantiated by the reference configs; unsupported')
        self.cfg = load_config(hrnet_config, head=head, upscale=upscale)
        self._L = _lib.lib()
        if dtype is None:
            dtype = self._L.sncal_x3_name().decode()
        self.dtype_name = dtype
        self.dtype = DTYPES[dtype]
        self.device = torch.device(device)
        if dtype in ('fp16x3', 'bf16x3') and dtype != self._L.sncal_x3_name().decode():
            raise _lib.SncalError(f"dtype {dtype!r}: this libsncal.so implements the split-arithmetic engine as "
                                  f"{self._L.sncal_x3_name().decode()!r} (rebuild with -DSNCAL_X3_F16={int(dtype == 'fp16x3')} for the other split type)")
        self._h = _lib.vp()
        desc = _desc(self.cfg)
        _lib.check(self._L.sncal_hrnet_create(ctypes.byref(desc), self.dtype, ctypes.byref(self._h)), 'sncal_hrnet_create')
        self.num_classes = desc.num_classes
        self._ws = None
        self._loaded = False
        self.equalize = True         # fp16x3: load-time rebalancing of block-internal channels (sncal_hrnet_set_equalize)
        self.equalized = 0

    def __del__(self):
        try:
            if getattr(self, '_h', None):
                self._L.sncal_hrnet_destroy(self._h)
                self._h = None
        except Exception:
            pass

    # ---- weights --------------------------------------------------------------------------------
    def conv_units(self):
        """[(name, bn_name, cin, cout, k, stride, has_bias)] in the reference's registration order."""
        out = []
        name = ctypes.create_string_buffer(128)
        bn = ctypes.create_string_buffer(128)
        ci, co, k, s, hb = (ctypes.c_int() for _ in range(5))
        for i in range(self._L.sncal_hrnet_num_convs(self._h)):
            _lib.check(self._L.sncal_hrnet_conv_info(self._h, i, name, 128, bn, 128, ctypes.byref(ci), ctypes.byref(co),
                                                     ctypes.byref(k), ctypes.byref(s), ctypes.byref(hb)), 'conv_info')
            out.append((name.value.decode(), bn.value.decode(), ci.value, co.value, k.value, s.value, bool(hb.value)))
        return out

    def load_state_dict(self, state_dict, strict: bool = True):
        """Accepts the reference's nn_state_dict (keys 'model.conv1.weight', ... metamodel.py:108-124)."""
        self.set_convs(state_dict, strict)
        with torch.cuda.device(self.device):
            _lib.check(self._L.sncal_hrnet_finalize(self._h), 'sncal_hrnet_finalize')
        self._loaded = True
        return self

    def set_convs(self, state_dict, strict: bool = True):
        """The host half of load_state_dict (no GPU needed): eval-mode BatchNorm folded in fp64, every conv unit handed to the library
        (sncal_hrnet_set_conv), and -- fp16x3 -- the library's power-of-two rebalancing of block-internal channels
        (sncal_hrnet_equalize; `equalized` = channels moved).  load_state_dict = set_convs + sncal_hrnet_finalize."""
        sd = {k.replace('_orig_mod.', ''): v for k, v in state_dict.items()}
        used = set()
        units = self.conv_units()
        for i, (name, bn, cin, cout, k, stride, has_bias) in enumerate(units):
            w = sd[name + '.weight'].detach().to('cpu', torch.float64)
            used.add(name + '.weight')
            if tuple(w.shape) != (cout, cin, k, k):
                raise _lib.SncalError(f'{name}.weight has shape {tuple(w.shape)}, expected {(cout, cin, k, k)}')
            b = torch.zeros(cout, dtype=torch.float64)
            if has_bias:
                b = sd[name + '.bias'].detach().to('cpu', torch.float64)
                used.add(name + '.bias')
            if bn:
                g = sd[bn + '.weight'].detach().to('cpu', torch.float64)
                beta = sd[bn + '.bias'].detach().to('cpu', torch.float64)
                mu = sd[bn + '.running_mean'].detach().to('cpu', torch.float64)
                var = sd[bn + '.running_var'].detach().to('cpu', torch.float64)
                used.update({bn + s for s in ('.weight', '.bias', '.running_mean', '.running_var')})
                scale = g / torch.sqrt(var + BN_EPS)
                shift = beta + (b - mu) * scale
            else:
                scale = torch.ones(cout, dtype=torch.float64)
                shift = b
            wf = np.ascontiguousarray(w.to(torch.float32).numpy())
            sc = np.ascontiguousarray(scale.to(torch.float32).numpy())
            sh = np.ascontiguousarray(shift.to(torch.float32).numpy())
            _lib.check(self._L.sncal_hrnet_set_conv(self._h, i, wf.ctypes.data, sc.ctypes.data, sh.ctypes.data), 'set_conv')
        if strict:
            extra = [k for k in sd if k not in used and not k.endswith('num_batches_tracked')]
            if extra:
                raise _lib.SncalError(f'unexpected keys in state dict: {extra[:5]}...')
        # fp16x3: block-internal channels are rebalanced by exact powers of two inside the library (sncal_hrnet_equalize = the first step
        # of sncal_hrnet_finalize; sncal.h "balance"), so that a caller of the C ABI gets what load_model gets
        _lib.check(self._L.sncal_hrnet_set_equalize(self._h, int(self.equalize)), 'sncal_hrnet_set_equalize')
        moved = ctypes.c_int()
        _lib.check(self._L.sncal_hrnet_equalize(self._h, ctypes.byref(moved)), 'sncal_hrnet_equalize')
        self.equalized = moved.value
        return self

    def folded_conv(self, idx):
        """(weight (cout,cin,k,k), scale (cout), shift (cout)) fp32 numpy of conv unit `idx` as the library holds them between
        sncal_hrnet_set_conv and sncal_hrnet_finalize (test instrumentation: the rebalanced parameters)."""
        name, bn, cin, cout, k, stride, has_bias = self.conv_units()[idx]
        w = np.empty((cout, cin, k, k), dtype=np.float32)
        sc, sh = np.empty(cout, dtype=np.float32), np.empty(cout, dtype=np.float32)
        _lib.check(self._L.sncal_hrnet_get_conv(self._h, idx, w.ctypes.data, sc.ctypes.data, sh.ctypes.data), 'sncal_hrnet_get_conv')
        return w, sc, sh

    # ---- forward --------------------------------------------------------------------------------
    def output_size(self, H, W):
        h, w = ctypes.c_int(), ctypes.c_int()
        _lib.check(self._L.sncal_hrnet_output_size(self._h, H, W, ctypes.byref(h), ctypes.byref(w)), 'output_size')
        return h.value, w.value

    def _workspace(self, B, H, W):
        n = ctypes.c_size_t()
        _lib.check(self._L.sncal_hrnet_workspace(self._h, B, H, W, ctypes.byref(n)), 'sncal_hrnet_workspace')
        if self._ws is None or self._ws.numel() < n.value:
            self._ws = torch.empty(n.value, dtype=torch.uint8, device=self.device)
        return self._ws

    def forward(self, x: torch.Tensor, want_heat: bool = True, decode_size=None):
        """x (B,3,H,W) fp32 in [0,1] on the GPU (ToTensor's output), or (B,H,W,3) uint8 (cv2.imread's frames, before
        ToTensor: identical results, a quarter of the input bytes).  Returns (heat or None, kpts or None)."""
        if not self._loaded:
            raise _lib.SncalError('HRNetHeatmap: load_state_dict() has not been called')
        u8 = x.dtype == torch.uint8
        if u8:
            _lib.require_device(x, torch.uint8, 'x')
            B, H, W, C = x.shape
        else:
            _lib.require_device(x, torch.float32, 'x')
            B, C, H, W = x.shape
        if C != 3:
            raise _lib.SncalError('x must have 3 channels')
        h, w = self.output_size(H, W)
        heat = torch.empty((B, self.num_classes, h, w), dtype=torch.float32, device=x.device) if want_heat else None
        kpts = None
        ih = iw = 0
        if decode_size is not None:
            ih, iw = int(decode_size[0]), int(decode_size[1])
            kpts = torch.empty((B, self.num_classes - 1, 3), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            ws = self._workspace(B, H, W)
            fn = self._L.sncal_hrnet_forward_u8 if u8 else self._L.sncal_hrnet_forward
            _lib.check(fn(self._h, x.data_ptr(), B, H, W,
                          heat.data_ptr() if heat is not None else None,
                          kpts.data_ptr() if kpts is not None else None, ih, iw,
                          ws.data_ptr(), ws.numel(), _lib.current_stream_ptr()),
                       'sncal_hrnet_forward')
        return heat, kpts

    def range_status(self, clear: bool = True, check: bool = False):
        """(overflow, nonfinite) since the last clear: wavefronts of the split-fp16 engine that clamped an activation beyond +-65504, and
        workgroups that met a NaN / infinite input value (sncal_hrnet_range_status; synchronises the current stream).  The reference's
        predict() is fp32 and has no such limit (metamodel.py:127-134): a non-zero count means the forwards since the last clear are NOT
        its result.  check=True raises SncalRangeError instead of returning the counts.  Always (0, 0) on the other engines."""
        ov, nf = ctypes.c_uint(), ctypes.c_uint()
        with torch.cuda.device(self.device):
            st = self._L.sncal_hrnet_range_status(self._h, ctypes.byref(ov), ctypes.byref(nf), int(bool(clear)), _lib.current_stream_ptr())
        if st != 0 and (check or st != _lib.ERR_RANGE):
            _lib.check(st, 'sncal_hrnet_range_status')
        return ov.value, nf.value

    # ---- C5: fp8 arithmetic in the wide 3x3 convolutions (dtype='fp8') -------------------------------------------
    def calibrate_fp8(self, x: torch.Tensor):
        """One bf16 forward of x (B,3,H,W) fp32 that records the per-tensor activation ranges the fp8 convolutions scale by."""
        _lib.require_device(x, torch.float32, 'x')
        B, C, H, W = x.shape
        with torch.cuda.device(x.device):
            ws = self._workspace(B, H, W)
            _lib.check(self._L.sncal_hrnet_calibrate_fp8(self._h, x.data_ptr(), B, H, W, ws.data_ptr(), ws.numel(),
                                                         _lib.current_stream_ptr()), 'sncal_hrnet_calibrate_fp8')
        self._ws = None            # the fp8 layout adds the e4m3 twins: re-query the workspace
        return self

    def set_fp8_layers(self, spec: str = 'all'):
        """Which wide 3x3 convolutions run in fp8: 'all', 'none', or e.g. 'stage4', 'stage3,stage4', 'c384', 'stage4,c192,c384'."""
        _lib.check(self._L.sncal_hrnet_set_fp8_layers(self._h, spec.encode()), 'sncal_hrnet_set_fp8_layers')
        self._ws = None
        return self

    def set_profiling(self, enable):
        """Per-launch HIP event timing (measurement only): False/0 off, True/1 every launch, 2 only the launches of
        the kernel variant that led the profile recorded so far."""
        _lib.check(self._L.sncal_hrnet_set_profiling(self._h, int(enable)), 'set_profiling')

    def get_profile(self):
        """[{kernel, flops, bytes, ms, launches}] accumulated since set_profiling(True) (synchronises)."""
        n = ctypes.c_int()
        buf = (_lib.KernelStat * 256)()
        _lib.check(self._L.sncal_hrnet_get_profile(self._h, buf, 256, ctypes.byref(n)), 'get_profile')
        return [dict(kernel=buf[i].kernel.decode(), flops=buf[i].flops, bytes=buf[i].bytes, ms=buf[i].ms,
                     launches=buf[i].launches) for i in range(min(n.value, 256))]

    # ---- plan introspection + taps (test instrumentation: tests/test_kernels_gpu.py) -----------------------------
    OP_TYPES = ('input', 'conv', 'upsample_add', 'softmax', 'decode', 'head')

    def plan_ops(self):
        """The executor's op list at the current layout (valid after a forward / workspace query): list of dicts."""
        out = []
        po = _lib.PlanOp()
        for i in range(self._L.sncal_hrnet_plan_num_ops(self._h)):
            _lib.check(self._L.sncal_hrnet_plan_op(self._h, i, ctypes.byref(po)), 'plan_op')
            out.append(dict(idx=i, type=self.OP_TYPES[po.type], active=bool(po.active), conv=po.conv, name=po.name.decode(),
                            cin=po.cin, cout=po.cout, ksize=po.ksize, stride=po.stride, col_off=po.col_off,
                            **{'in': po.in_}, res=po.res, out=po.out, base=po.base, src=list(po.src)[:po.nsrc],
                            head_direct=po.head_direct, head_src=list(po.head_src)[:po.head_nsrc],
                            head_fold=list(po.head_fold)[:po.head_nfold], relu=bool(po.relu), out_coff=po.out_coff,
                            out_f32=bool(po.out_f32), fp8=po.fp8 == 1, x3=po.fp8 == 2, x3g=po.fp8 == 3, res_twin=bool(po.res_twin), kernel=po.kernel.decode()))
        return out

    def plan_tensor(self, tid):
        pt = _lib.PlanTensor()
        _lib.check(self._L.sncal_hrnet_plan_tensor(self._h, tid, ctypes.byref(pt)), 'plan_tensor')
        return dict(id=tid, C=pt.C, H=pt.H, W=pt.W, dtype=('fp32', 'bf16', 'e4m3')[pt.dtype], twin=pt.twin, alive=bool(pt.alive),
                    scale=pt.scale, bytes=pt.bytes, sub_batch=pt.sub_batch)

    def clear_taps(self):
        _lib.check(self._L.sncal_hrnet_plan_tap(self._h, -1, 0, None), 'plan_tap')
        self._taps = []

    def tap(self, op_idx, tid):
        """Ask the next forward(s) to copy tensor `tid` (first sub-batch, NHWC) when the executor passes op `op_idx`.
        Returns the destination tensor (sub_batch, H, W, C) in the tensor's storage type."""
        info = self.plan_tensor(tid)
        tdt = {'fp32': torch.float32, 'bf16': torch.bfloat16, 'e4m3': torch.uint8}[info['dtype']]
        dst = torch.empty((info['sub_batch'], info['H'], info['W'], info['C']), dtype=tdt, device=self.device)
        assert dst.numel() * dst.element_size() == info['bytes'], info
        _lib.check(self._L.sncal_hrnet_plan_tap(self._h, op_idx, tid, dst.data_ptr()), 'plan_tap')
        self._taps = getattr(self, '_taps', []) + [dst]          # keep the buffers alive while the taps are registered
        return dst

    def __call__(self, x):
        """Reference nn.Module contract: list of stage outputs, [-1] is the head output."""
        return [self.forward(x, want_heat=True)[0]]

    def eval(self):
        return self

    def to(self, device):
        if torch.device(device) != self.device:
            raise _lib.SncalError('weights live on the device given at construction')
        return self
