"""ctypes binding of libsncal.so (the C ABI declared in include/sncal.h).

There is NO fallback: if the library is missing or a symbol cannot be resolved, importing the product
path raises.  PyTorch is used by the callers only for device memory and streams.
"""
import ctypes
import os

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('SNCAL_LIB_PATH') or os.path.join(_PKG, 'libsncal.so')     # override: A/B runs of two builds

c_float_p = ctypes.POINTER(ctypes.c_float)
c_double_p = ctypes.POINTER(ctypes.c_double)
c_int_p = ctypes.POINTER(ctypes.c_int)
vp = ctypes.c_void_p


class HRNetDesc(ctypes.Structure):
    """sncal_hrnet_desc."""
    _fields_ = [('num_classes', ctypes.c_int), ('stem_width', ctypes.c_int), ('upscale', ctypes.c_int),
                ('head_softmax', ctypes.c_int), ('stage1_blocks', ctypes.c_int),
                ('stage1_channels', ctypes.c_int), ('num_modules', ctypes.c_int * 3),
                ('num_branches', ctypes.c_int * 3), ('num_blocks', ctypes.c_int * 3),
                ('num_channels', (ctypes.c_int * 4) * 3)]


class KernelStat(ctypes.Structure):
    """sncal_kernel_stat."""
    _fields_ = [('kernel', ctypes.c_char * 96), ('flops', ctypes.c_double), ('bytes', ctypes.c_double),
                ('ms', ctypes.c_double), ('launches', ctypes.c_int)]


class PlanOp(ctypes.Structure):
    """sncal_plan_op."""
    _fields_ = [('type', ctypes.c_int), ('active', ctypes.c_int), ('conv', ctypes.c_int), ('name', ctypes.c_char * 96),
                ('cin', ctypes.c_int), ('cout', ctypes.c_int), ('ksize', ctypes.c_int), ('stride', ctypes.c_int),
                ('col_off', ctypes.c_int), ('in_', ctypes.c_int), ('res', ctypes.c_int), ('out', ctypes.c_int),
                ('base', ctypes.c_int), ('src', ctypes.c_int * 4), ('nsrc', ctypes.c_int), ('head_direct', ctypes.c_int),
                ('head_src', ctypes.c_int * 5), ('head_nsrc', ctypes.c_int), ('head_fold', ctypes.c_int * 2),
                ('head_nfold', ctypes.c_int), ('relu', ctypes.c_int), ('out_coff', ctypes.c_int), ('out_f32', ctypes.c_int),
                ('fp8', ctypes.c_int), ('kernel', ctypes.c_char * 96), ('res_twin', ctypes.c_int)]


class PlanTensor(ctypes.Structure):
    """sncal_plan_tensor."""
    _fields_ = [('C', ctypes.c_int), ('H', ctypes.c_int), ('W', ctypes.c_int), ('dtype', ctypes.c_int), ('twin', ctypes.c_int),
                ('alive', ctypes.c_int), ('scale', ctypes.c_float), ('bytes', ctypes.c_size_t), ('sub_batch', ctypes.c_int)]


class Camera(ctypes.Structure):
    """sncal_camera."""
    _fields_ = [('position', ctypes.c_double * 3), ('rotation', ctypes.c_double * 9),
                ('fx', ctypes.c_double), ('fy', ctypes.c_double), ('cx', ctypes.c_double),
                ('cy', ctypes.c_double), ('rmse', ctypes.c_double), ('status', ctypes.c_int32),
                ('n_points', ctypes.c_int32)]


MAX_CONF_THRESHS = 16          # SNCAL_MAX_CONF_THRESHS


class VoterCfg(ctypes.Structure):
    """sncal_voter_cfg."""
    _fields_ = [('algorithm', ctypes.c_int), ('n_conf_threshs', ctypes.c_int), ('conf_thresh', ctypes.c_double),
                ('conf_threshs', ctypes.c_double * MAX_CONF_THRESHS),
                ('max_rmse', ctypes.c_double), ('max_rmse_rel', ctypes.c_double),
                ('min_points', ctypes.c_int), ('min_points_per_plane', ctypes.c_int),
                ('min_points_for_refinement', ctypes.c_int), ('reliable_thresh', ctypes.c_int),
                ('min_focal_length', ctypes.c_double), ('img_w', ctypes.c_int), ('img_h', ctypes.c_int),
                ('lm_schedule', ctypes.c_int), ('refine_max_iters', ctypes.c_int)]


class JpegInfo(ctypes.Structure):
    """sncal_jpeg_info."""
    _fields_ = [('width', ctypes.c_int32), ('height', ctypes.c_int32), ('components', ctypes.c_int32),
                ('h_samp', ctypes.c_int32), ('v_samp', ctypes.c_int32), ('restart_interval', ctypes.c_int32),
                ('blocks', ctypes.c_int32 * 3)]


# name -> (restype, argtypes); must list every function include/sncal.h declares
SIGNATURES = {
    'sncal_version': (ctypes.c_int, []),
    'sncal_last_error': (ctypes.c_char_p, []),
    'sncal_x3_name': (ctypes.c_char_p, []),
    'sncal_heatmap_decode': (ctypes.c_int, [vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                            ctypes.c_int, ctypes.c_int, vp, vp]),
    'sncal_line_decode': (ctypes.c_int, [vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                         ctypes.c_float, ctypes.c_float, vp, vp]),
    'sncal_hrnet_create': (ctypes.c_int, [ctypes.POINTER(HRNetDesc), ctypes.c_int, ctypes.POINTER(vp)]),
    'sncal_hrnet_destroy': (None, [vp]),
    'sncal_hrnet_num_convs': (ctypes.c_int, [vp]),
    'sncal_hrnet_conv_info': (ctypes.c_int, [vp, ctypes.c_int, ctypes.c_char_p, ctypes.c_int, ctypes.c_char_p,
                                             ctypes.c_int, c_int_p, c_int_p, c_int_p, c_int_p, c_int_p]),
    'sncal_hrnet_set_conv': (ctypes.c_int, [vp, ctypes.c_int, vp, vp, vp]),
    'sncal_hrnet_finalize': (ctypes.c_int, [vp]),
    'sncal_hrnet_set_equalize': (ctypes.c_int, [vp, ctypes.c_int]),
    'sncal_hrnet_equalize': (ctypes.c_int, [vp, c_int_p]),
    'sncal_hrnet_get_conv': (ctypes.c_int, [vp, ctypes.c_int, vp, vp, vp]),
    'sncal_hrnet_range_status': (ctypes.c_int, [vp, ctypes.POINTER(ctypes.c_uint), ctypes.POINTER(ctypes.c_uint), ctypes.c_int, vp]),
    'sncal_hrnet_output_size': (ctypes.c_int, [vp, ctypes.c_int, ctypes.c_int, c_int_p, c_int_p]),
    'sncal_hrnet_workspace': (ctypes.c_int, [vp, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                             ctypes.POINTER(ctypes.c_size_t)]),
    'sncal_hrnet_forward': (ctypes.c_int, [vp, vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, vp,
                                           ctypes.c_int, ctypes.c_int, vp, ctypes.c_size_t, vp]),
    'sncal_evaluate_cameras': (ctypes.c_int, [vp, ctypes.c_int, vp, vp, vp, ctypes.c_int, vp, vp, vp, ctypes.c_int,
                                              ctypes.c_double, ctypes.c_int, ctypes.c_int, vp, vp]),
    'sncal_evaluate_cameras_detail': (ctypes.c_int, [vp, ctypes.c_int, vp, vp, vp, ctypes.c_int, vp, vp, vp, ctypes.c_int,
                                                     ctypes.c_double, ctypes.c_int, ctypes.c_int, vp, vp, vp, vp]),
    'sncal_lines_to_points': (ctypes.c_int, [vp, ctypes.c_int, ctypes.c_float, ctypes.c_double, vp, vp]),
    'sncal_hrnet_forward_u8': (ctypes.c_int, [vp, vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, vp,
                                              ctypes.c_int, ctypes.c_int, vp, ctypes.c_size_t, vp]),
    'sncal_hrnet_set_profiling': (ctypes.c_int, [vp, ctypes.c_int]),
    'sncal_hrnet_calibrate_fp8': (ctypes.c_int, [vp, vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, ctypes.c_size_t, vp]),
    'sncal_hrnet_set_fp8_layers': (ctypes.c_int, [vp, ctypes.c_char_p]),
    'sncal_hrnet_plan_num_ops': (ctypes.c_int, [vp]),
    'sncal_hrnet_plan_num_tensors': (ctypes.c_int, [vp]),
    'sncal_hrnet_plan_op': (ctypes.c_int, [vp, ctypes.c_int, ctypes.POINTER(PlanOp)]),
    'sncal_hrnet_plan_tensor': (ctypes.c_int, [vp, ctypes.c_int, ctypes.POINTER(PlanTensor)]),
    'sncal_hrnet_plan_tap': (ctypes.c_int, [vp, ctypes.c_int, ctypes.c_int, vp]),
    'sncal_hrnet_get_profile': (ctypes.c_int, [vp, ctypes.POINTER(KernelStat), ctypes.c_int, c_int_p]),
    'sncal_pnp_refine_lm': (ctypes.c_int, [vp, vp, vp, vp, ctypes.c_int, ctypes.c_int, vp, vp, ctypes.c_int,
                                           ctypes.c_double, vp]),
    'sncal_solve_pnp': (ctypes.c_int, [vp, vp, vp, vp, ctypes.c_int, ctypes.c_int, vp, vp]),
    'sncal_jpeg_probe': (ctypes.c_int, [vp, ctypes.c_size_t, ctypes.POINTER(JpegInfo)]),
    'sncal_jpeg_entropy_decode': (ctypes.c_int, [vp, ctypes.c_size_t, vp, ctypes.c_size_t, ctypes.POINTER(JpegInfo)]),
    'sncal_jpeg_create': (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(vp)]),
    'sncal_jpeg_destroy': (None, [vp]),
    'sncal_jpeg_decode': (ctypes.c_int, [vp, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(ctypes.c_size_t),
                                         ctypes.c_int, vp, vp]),
    'sncal_create_target': (ctypes.c_int, [vp, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_int, ctypes.c_int, vp, vp]),
    'sncal_calibrate': (ctypes.c_int, [vp, vp, ctypes.c_int, ctypes.POINTER(VoterCfg), vp, vp]),
    'sncal_calibrate_workspace': (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(VoterCfg), ctypes.POINTER(ctypes.c_size_t)]),
    'sncal_calibrate_ws': (ctypes.c_int, [vp, vp, ctypes.c_int, ctypes.POINTER(VoterCfg), vp, vp, ctypes.c_size_t, vp]),
    'sncal_shutdown': (ctypes.c_int, []),
    'sncal_stream_create_cu_mask': (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(vp)]),
    'sncal_stream_destroy': (ctypes.c_int, [vp]),
}

_lib = None
MISSING = []


class SncalError(RuntimeError):
    pass


def lib():
    """Load libsncal.so once; raise loudly if it is absent (no CPU fallback exists)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SncalError(f'{LIB_PATH} is missing: build it with __graft_entry__.build() '
                             '(hipcc --offload-arch=gfx950); this package has no CPU fallback')
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            try:
                fn = getattr(handle, name)
            except AttributeError:          # calling it later raises; tests/test_abi.py requires none
                MISSING.append(name)
                continue
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


class SncalRangeError(SncalError):
    """SNCAL_ERR_RANGE: the fp16x3 engine met a folded weight (load_state_dict) or an activation (range_status) that fp16 hi + lo
    halves do not represent; the reference's fp32 predict() has no such limit -- use dtype='fp32'."""


ERR_RANGE = -6


def check(status, what=''):
    if status != 0:
        msg = lib().sncal_last_error().decode(errors='replace')
        raise (SncalRangeError if status == ERR_RANGE else SncalError)(f'{what} failed with status {status}: {msg}')


def current_stream_ptr():
    """hipStream_t of torch's current stream as an integer (0 = default stream)."""
    import torch
    return int(torch.cuda.current_stream().cuda_stream)


def require_device(t, dtype, name):
    import torch
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise SncalError(f'{name} must be a tensor on the GPU (libsncal has no CPU path)')
    if t.dtype != dtype:
        raise SncalError(f'{name} must be {dtype}, got {t.dtype}')
    if not t.is_contiguous():
        raise SncalError(f'{name} must be contiguous')
    return t
