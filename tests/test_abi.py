"""CPU: libsncal.so loads and exports every function include/sncal.h declares (no compute calls)."""
import ctypes
import os
import re

import numpy as np
import pytest

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
    assert ctypes.sizeof(L.VoterCfg) == 8 + 8 + 16 * 8 + 16 + 4 * 4 + 8 + 8 + 8  # sncal_voter_cfg (img_w, img_h, lm_schedule, refine_max_iters last)


def test_struct_layouts_match_what_a_c_compiler_sees(tmp_path):
    """sizeof / offsetof of every struct of include/sncal.h as gcc lays it out == the ctypes mirrors, field by field; and the
    VoterCfg binding INTEGRATION.md shows a maintainer == the header's struct (the snippet was two fields short)."""
    import re
    import shutil
    import subprocess
    import sncal_amd
    L = sncal_amd._lib
    gcc = shutil.which('gcc')
    if gcc is None:
        pytest.skip('no gcc')
    structs = {'sncal_voter_cfg': L.VoterCfg, 'sncal_camera': L.Camera, 'sncal_hrnet_desc': L.HRNetDesc, 'sncal_kernel_stat': L.KernelStat,
               'sncal_plan_op': L.PlanOp, 'sncal_plan_tensor': L.PlanTensor}
    rename = {'in_': 'in'}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "' + os.path.join(ROOT, 'include', 'sncal.h') + '"', 'int main(void) {']
    for cname, ct in structs.items():
        lines.append(f'  printf("{cname} %zu\\n", sizeof({cname}));')
        for f in ct._fields_:
            fn = rename.get(f[0], f[0])
            lines.append(f'  printf("{cname}.{f[0]} %zu\\n", offsetof({cname}, {fn}));')
    lines += ['  return 0;', '}']
    src = tmp_path / 'layout.c'
    src.write_text('\n'.join(lines))
    exe = tmp_path / 'layout'
    subprocess.check_call([gcc, str(src), '-o', str(exe)])
    seen = dict(ln.split() for ln in subprocess.check_output([str(exe)]).decode().splitlines())
    for cname, ct in structs.items():
        assert int(seen[cname]) == ctypes.sizeof(ct), cname
        for f in ct._fields_:
            assert int(seen[f'{cname}.{f[0]}']) == getattr(ct, f[0]).offset, (cname, f[0])
    # the binding INTEGRATION.md shows
    text = open(os.path.join(ROOT, 'INTEGRATION.md')).read()
    m = re.search(r'class VoterCfg\(ctypes\.Structure\):.*?\n(    _fields_ = \[.*?\]\s*(?:#[^\n]*)?)\n(?=cfg = )', text, re.S)
    assert m, 'INTEGRATION.md no longer shows the VoterCfg binding'
    ns = {'ctypes': ctypes}
    exec('class VoterCfg(ctypes.Structure):\n' + m.group(1), ns)
    doc = ns['VoterCfg']
    assert [f[0] for f in doc._fields_] == [f[0] for f in L.VoterCfg._fields_]
    assert ctypes.sizeof(doc) == ctypes.sizeof(L.VoterCfg)
    assert all(getattr(doc, f[0]).offset == getattr(L.VoterCfg, f[0]).offset for f in doc._fields_)


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


def test_file_interop_camera_json_and_lines_pickle(gold_dir, tmp_path):
    """N1: camera_<frame>.json byte-compatible with what the reference's PredictionSaver writes for the same
    camera (golden JSON dicts captured from the reference Camera), and the lines pickle in both layouts."""
    import json
    import sncal_amd
    from sncal_amd import interop
    g = np.load(os.path.join(gold_dir, 'camera.npz'), allow_pickle=True)
    names = []
    cams = []
    for i in range(int(g['n'])):
        ref = json.loads(str(g[f'{i}.json']))
        cam = sncal_amd.Camera(960, 540)
        cam.from_json_parameters(ref)
        cams.append(cam if i % 2 == 0 else None)
        names.append(f'{i:05d}.jpg')
        if i % 2 == 0:
            p = interop.camera_json_path(str(tmp_path), names[-1])
            interop.save_camera_json(cam, p)
            assert os.path.basename(p) == f'camera_{i:05d}.json'
            # same serialisation call as PredictionSaver (json.dump(..., indent=4)) on the mirrored to_json_parameters(),
            # whose values equal the reference's for the same camera (tests/test_oracle_goldens.py pins them)
            assert open(p).read() == json.dumps(cam.to_json_parameters(), indent=4)
            written = json.load(open(p))
            assert list(written) == list(ref)                         # same keys, same order
            for k in ref:
                assert np.allclose(written[k], ref[k], rtol=1e-9, atol=1e-9), k
            rt = interop.load_camera_json(p)
            assert np.allclose(rt.position, cam.position) and np.allclose(rt.rotation, cam.rotation, atol=1e-12)
    assert interop.save_cameras(cams, names, str(tmp_path / 'out')) == sum(c is not None for c in cams)
    per_image = {'a.jpg': {'lines': {'Middle line': (0.5, 12.0), 'Side line top': (-0.01, 40.0)},
                           'points': {'Middle line': [(10.0, 17.0, 0.9), (30.0, 27.0, 0.8)]}}}
    for as_list in (True, False):
        path = str(tmp_path / f'lines_{as_list}.pkl')
        interop.save_lines_pickle(per_image, path, as_list=as_list)
        assert interop.load_lines_pickle(path) == per_image
    cc = sncal_amd.CameraCreator(sncal_amd.PITCH_POINTS, lines_file=str(tmp_path / 'lines_True.pkl'))
    assert 15 in cc.lines_data['a.jpg']          # Middle line x Side line top (intersections.py:28)


def test_jpeg_host_stage_matches_oracle_and_refuses_what_it_cannot_decode():
    """N3 host half (no GPU): header parse + Huffman decode of libsncal equal the oracle's coefficient blocks on
    every golden stream; unsupported / damaged input fails with the documented status and a message."""
    import sncal_amd
    from oracle import jpeg as oj
    L = sncal_amd._lib
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'jpeg_cases.npz'))
    names = [str(n) for n in g['names']] + ['full']
    for n in names:
        data = g['jpg.' + n].tobytes()
        frame = oj.parse(data)
        info = sncal_amd.jpeg.probe(data)
        assert (info['width'], info['height'], info['components']) == (frame['width'], frame['height'], len(frame['comps']))
        assert (info['h_samp'], info['v_samp']) == (frame['comps'][0]['h'], frame['comps'][0]['v'])
        assert info['restart_interval'] == frame['restart']
        want = oj.decode_coefficients(frame)
        got = sncal_amd.jpeg.entropy_decode(data)
        assert len(want) == len(got)
        for a, b in zip(want, got):
            assert a.shape == b.shape and np.array_equal(a, b), n

    def status(data):
        info = L.JpegInfo()
        return L.lib().sncal_jpeg_probe(data, len(data), ctypes.byref(info))

    assert status(g['jpg.progressive'].tobytes()) == -5                 # SNCAL_ERR_UNSUPPORTED
    assert b'SOF2' in L.lib().sncal_last_error()
    good = g['jpg.full'].tobytes()
    assert status(good[:200]) == -1                                       # truncated inside the tables
    assert status(b'\x89PNG\r\n\x1a\n' + good[8:]) == -1                  # not a JPEG
    assert status(b'') == -1
    # EXIF orientation: cv2.imread rotates / flips such frames, the decoder refuses them (orientation 1 = upright is fine)
    import struct

    def with_exif(data, orientation, little=True):
        e = '<' if little else '>'
        tiff = (b'II*\x00' if little else b'MM\x00*') + struct.pack(e + 'I', 8) + struct.pack(e + 'H', 1) + \
            struct.pack(e + 'HHI', 0x0112, 3, 1) + struct.pack(e + 'HH', orientation, 0) + struct.pack(e + 'I', 0)
        seg = b'Exif\x00\x00' + tiff
        return data[:2] + b'\xff\xe1' + struct.pack('>H', len(seg) + 2) + seg + data[2:]
    assert status(with_exif(good, 1)) == 0 and status(with_exif(good, 1, little=False)) == 0
    for o in (3, 6, 8):
        assert status(with_exif(good, o)) == -5 and b'EXIF orientation' in L.lib().sncal_last_error()
        assert status(with_exif(good, o, little=False)) == -5
    # capacity check of the coefficient buffer
    info = L.JpegInfo()
    buf = np.zeros(64, np.int16)
    assert L.lib().sncal_jpeg_entropy_decode(good, len(good), buf.ctypes.data, 64, ctypes.byref(info)) == -4
    # a scan whose Huffman data is damaged: a code no table defines
    sos = good.index(b'\xff\xda')
    hdr_len = (good[sos + 2] << 8) | good[sos + 3]
    broken = good[:sos + 2 + hdr_len] + b'\xff\x00' * 4000 + b'\xff\xd9'
    big = np.zeros(sum(sncal_amd.jpeg.probe(good)['blocks']) * 64, np.int16)
    st = L.lib().sncal_jpeg_entropy_decode(broken, len(broken), big.ctypes.data, big.size, ctypes.byref(info))
    assert st == -1 and b'MCU' in L.lib().sncal_last_error()


def test_argument_validation_and_empty_batches_without_a_gpu():
    """Every compute entry point validates its arguments before touching the device and accepts an empty batch:
    statuses and messages as include/sncal.h documents (no kernel is launched here)."""
    import sncal_amd
    L = sncal_amd._lib
    lib = L.lib()
    ERR_ARG = -1
    one = ctypes.c_void_p(16)                     # a non-null dummy pointer; never dereferenced on these paths
    cfg = L.VoterCfg()
    cfg.algorithm, cfg.n_conf_threshs, cfg.img_w, cfg.img_h = 0, 3, 960, 540
    # empty batches
    assert lib.sncal_heatmap_decode(None, 0, 58, 270, 480, 540, 960, None, None) == 0
    assert lib.sncal_line_decode(None, 0, 23, 135, 240, 3.0, 4.0, None, None) == 0
    assert lib.sncal_lines_to_points(None, 0, 4.0, 0.0, None, None) == 0
    assert lib.sncal_calibrate(None, None, 0, ctypes.byref(cfg), None, None) == 0
    assert lib.sncal_calibrate_ws(None, None, 0, ctypes.byref(cfg), None, None, 0, None) == 0
    n1, n64 = ctypes.c_size_t(), ctypes.c_size_t()                      # the solve's caller-owned workspace: a host-only size query
    assert lib.sncal_calibrate_workspace(1, ctypes.byref(cfg), ctypes.byref(n1)) == 0 and lib.sncal_calibrate_workspace(64, ctypes.byref(cfg), ctypes.byref(n64)) == 0
    assert 256 <= n1.value < n64.value <= 64 * 3 * 2048 + 64 * 1024
    assert lib.sncal_calibrate_workspace(64, None, ctypes.byref(n64)) == ERR_ARG
    assert lib.sncal_shutdown() == 0                                    # nothing held: a no-op that must not need a GPU
    assert lib.sncal_create_target(None, 0, 57, 3.0, 270, 480, None, None) == 0
    assert lib.sncal_evaluate_cameras(None, 0, None, None, None, 26, None, None, None, 4, 5.0, 960, 540, None, None) == 0
    # bad shapes / null pointers
    assert lib.sncal_heatmap_decode(one, 1, 1, 270, 480, 540, 960, one, None) == ERR_ARG          # C < 2
    assert lib.sncal_heatmap_decode(None, 1, 58, 270, 480, 540, 960, one, None) == ERR_ARG and b'null' in lib.sncal_last_error()
    assert lib.sncal_line_decode(one, 1, 23, 135, 240, 0.0, 4.0, one, None) == ERR_ARG             # sigma <= 0
    assert lib.sncal_calibrate(None, None, 1, ctypes.byref(cfg), one, None) == ERR_ARG
    cfg.algorithm = 9
    assert lib.sncal_calibrate(one, None, 1, ctypes.byref(cfg), one, None) == ERR_ARG and b'algorithm' in lib.sncal_last_error()
    assert lib.sncal_create_target(one, 1, 65, 3.0, 8, 8, one, None) == ERR_ARG
    assert lib.sncal_create_target(one, 1, 57, -1.0, 8, 8, one, None) == ERR_ARG
    assert lib.sncal_evaluate_cameras(one, 1, one, one, one, 64, one, one, one, 4, 5.0, 960, 540, one, None) == ERR_ARG   # n_cls > 32
    assert lib.sncal_jpeg_create(0, 540, 960, 1, ctypes.byref(ctypes.c_void_p())) == ERR_ARG
    assert lib.sncal_jpeg_decode(None, None, None, 1, None, None) == ERR_ARG
    desc = L.HRNetDesc()
    h = ctypes.c_void_p()
    assert lib.sncal_hrnet_create(ctypes.byref(desc), 1, ctypes.byref(h)) == ERR_ARG               # zeroed descriptor
    assert lib.sncal_hrnet_forward(None, None, 1, 540, 960, None, None, 540, 960, None, 0, None) == ERR_ARG


def test_jpeg_host_parser_survives_damaged_streams():
    """The JPEG host stage parses untrusted bytes: 600 mutations of golden streams (bit flips, truncations, spliced
    segments, oversized dimensions) must each come back with a status -- OK or a documented error -- and never read or
    write out of bounds (the coefficient buffer carries guard words)."""
    import sncal_amd
    L = sncal_amd._lib
    lib = L.lib()
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'jpeg_cases.npz'))
    names = [str(n) for n in g['names']]
    rng = np.random.default_rng(11)
    seen = set()
    for it in range(600):
        src = bytearray(g['jpg.' + names[int(rng.integers(len(names)))]].tobytes())
        kind = it % 4
        if kind == 0:                                        # random bit flips anywhere (headers included)
            for _ in range(int(rng.integers(1, 6))):
                src[int(rng.integers(len(src)))] ^= 1 << int(rng.integers(8))
        elif kind == 1:                                      # truncation
            src = src[:int(rng.integers(2, len(src)))]
        elif kind == 2:                                      # a chunk of the stream overwritten by another part of it
            a, b, n = (int(rng.integers(len(src))) for _ in range(3))
            n = min(n % 64 + 1, len(src) - max(a, b))
            src[a:a + n] = src[b:b + n]
        else:                                                # frame header: dimensions / sampling factors / component count
            sof = bytes(src).find(b'\xff\xc0')
            if sof >= 0:
                src[sof + 5 + int(rng.integers(0, 12))] = int(rng.integers(256))
        data = bytes(src)
        info = L.JpegInfo()
        st = lib.sncal_jpeg_probe(data, len(data), ctypes.byref(info))
        assert st in (0, -1, -5), st
        cap = 1 << 16                                        # small on purpose: big frames must be refused with -4
        buf = np.full(cap + 64, 0x5A5A, np.int16)
        st2 = lib.sncal_jpeg_entropy_decode(data, len(data), buf.ctypes.data, cap, ctypes.byref(info))
        assert st2 in (0, -1, -4, -5), st2
        assert (buf[cap:] == 0x5A5A).all()
        seen.add(st2)
    assert {0, -1} <= seen


def test_config_mappings_that_are_not_dicts():
    """Checkpoints written under hydra hold OmegaConf nodes: Mappings / Sequences that are not dict / list.  load_config
    and the meta-model constructor must take them; class-level head/upscale override whatever the yaml says."""
    import collections.abc
    import pytest
    import sncal_amd
    from sncal_amd import hrnet as H

    class Node(collections.abc.Mapping):
        def __init__(self, d): self._d = d
        def __getitem__(self, k):
            v = self._d[k]
            return Node(v) if isinstance(v, dict) else Seq(v) if isinstance(v, list) else v
        def __iter__(self): return iter(self._d)
        def __len__(self): return len(self._d)

    class Seq(collections.abc.Sequence):
        def __init__(self, v): self._v = v
        def __getitem__(self, i): return self._v[i]
        def __len__(self): return len(self._v)

    base = H.load_config('line_hrnet_w48')
    bare = {k: v for k, v in base.items() if k not in ('head', 'upscale')}
    cfg = H.load_config(Node(bare), head='softmax', upscale=1)
    assert type(cfg) is dict and type(cfg['stage2']) is dict and type(cfg['stage2']['num_channels']) is list
    assert cfg['head'] == 'softmax' and cfg['upscale'] == 1 and cfg['stage4']['num_channels'] == [48, 96, 192, 384]
    assert H.load_config(Node(bare))['head'] == 'logsoftmax'           # the keypoint class reads the yaml / default
    d = H._desc(cfg)
    assert d.head_softmax == 1 and d.upscale == 1 and d.num_classes == 23
    with pytest.raises(sncal_amd._lib.SncalError):
        H.load_config(12345)
    with pytest.raises(sncal_amd._lib.SncalError):
        H.load_config('no_such_config_name')
    with pytest.raises(sncal_amd._lib.SncalError):                       # more thresholds than the C struct carries
        sncal_amd.CameraCreator(sncal_amd.PITCH_POINTS, conf_threshs=[0.9 - 0.05 * i for i in range(17)])._cfg()
    c5 = sncal_amd.CameraCreator(sncal_amd.PITCH_POINTS, conf_threshs=[0.5, 0.4, 0.3, 0.2, 0.1])._cfg()     # the reference loops over any number
    assert c5.n_conf_threshs == 5 and list(c5.conf_threshs)[:5] == [0.5, 0.4, 0.3, 0.2, 0.1]
