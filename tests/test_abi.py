"""CPU: libsncal.so loads and exports every function include/sncal.h declares (no compute calls)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, 'include', 'sncal.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(sncal_[a-z0-9_]+)\s*\(', src)))


def test_library_exports_every_declared_symbol():
    import sncal_amd
    names = _declared()
    assert len(names) >= 16
    lib = ctypes.CDLL(sncal_amd._lib.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), f'{n} declared in include/sncal.h but not exported'
    assert lib.sncal_version() == 100


def test_python_binding_covers_the_header():
    import sncal_amd
    sncal_amd._lib.lib()
    assert sncal_amd._lib.MISSING == []
    assert sorted(sncal_amd._lib.SIGNATURES) == _declared()


def test_struct_layouts_match_the_header():
    import sncal_amd
    L = sncal_amd._lib
    assert ctypes.sizeof(L.Camera) == 3 * 8 + 9 * 8 + 5 * 8 + 2 * 4          # sncal_camera
    assert ctypes.sizeof(L.HRNetDesc) == (6 + 3 + 3 + 3 + 12) * 4            # sncal_hrnet_desc
    assert ctypes.sizeof(L.VoterCfg) == 8 + 8 + 32 + 16 + 4 * 4 + 8 + 8      # sncal_voter_cfg


def test_plan_enumeration_matches_the_oracle_without_a_gpu():
    import sncal_amd
    from oracle import hrnet_ref as hr
    for cfg in ('hrnet_w48', 'hrnet_w18', 'hrnet_w32', 'line_hrnet_w48'):
        net = sncal_amd.HRNetHeatmap(cfg, dtype='bf16', device='cpu')
        mine = [(u[0], u[1], u[2], u[3], u[4], u[5], u[6]) for u in net.conv_units()]
        ref = [(o.name, o.bn or '', o.cin, o.cout, o.k, o.stride, o.bias) for o in hr.enumerate_convs(hr.load_config(cfg))]
        assert mine == ref
    assert sncal_amd.HRNetHeatmap('hrnet_w48', device='cpu').output_size(540, 960) == (270, 480)


def test_product_path_fails_loudly_without_gpu_tensors():
    import pytest
    import torch
    import sncal_amd
    with pytest.raises(sncal_amd._lib.SncalError):
        sncal_amd.HRNetPredictionTransform((540, 960))(torch.zeros(1, 58, 8, 8))
    if not torch.cuda.is_available():
        with pytest.raises(sncal_amd._lib.SncalError):
            sncal_amd.CameraCreator(sncal_amd.PITCH_POINTS, algorithm='voter').solve_batch(torch.zeros(1, 57, 3))
