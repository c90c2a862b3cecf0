"""GPU: JPEG input stage through the C ABI vs the libjpeg-turbo capture and the oracle, bit-exact."""
import io
import os

import numpy as np
import pytest
import torch

from oracle import hrnet_ref as hr
from oracle import jpeg as oj

pytestmark = pytest.mark.gpu


def _cases(gold_dir):
    g = np.load(os.path.join(gold_dir, 'jpeg_cases.npz'))
    return g, [str(n) for n in g['names']]


def test_decoded_frames_equal_libjpeg_turbo_bit_exact(sncal, cuda, gold_dir):
    """Every golden stream (4:4:4 / 4:2:2 / 4:2:0 / grey, odd sizes down to 1x1, restart intervals, three
    qualities), batched by frame size so that one launch mixes sampling layouts."""
    g, names = _cases(gold_dir)
    by_size = {}
    for n in names:
        by_size.setdefault(n.split('_')[0], []).append(n)
    for size, group in by_size.items():
        h, w = (int(v) for v in size.split('x'))
        dec = sncal.JpegDecoder(h, w, max_batch=len(group), threads=4, device=cuda)
        out = dec.decode([g['jpg.' + n].tobytes() for n in group]).cpu().numpy()
        for i, n in enumerate(group):
            assert np.array_equal(out[i], g['bgr.' + n]), n
        dec.close()


def test_full_size_batch_matches_oracle_and_feeds_the_network(sncal, cuda, gold_dir):
    """960x540 (config C3): a batch with the golden frame repeated between re-encodings in other layouts; decoded
    frames equal the oracle's, repeated calls reuse both staging slots, and the frames drive sncal_hrnet_forward_u8."""
    g, _ = _cases(gold_dir)
    full = g['jpg.full'].tobytes()
    want = oj.decode_bgr(full)
    assert np.array_equal(want.astype(np.int64).sum(axis=(1, 2)), g['bgr.full.rowsum'])
    blobs = [full]
    try:
        from PIL import Image
        rgb = np.ascontiguousarray(want[..., ::-1])
        for kw in (dict(quality=90, subsampling=0), dict(quality=70, subsampling=1), dict(quality=50, subsampling=2, restart_marker_blocks=60)):
            b = io.BytesIO()
            Image.fromarray(rgb).save(b, 'JPEG', **kw)
            blobs.append(b.getvalue())
        pil = [np.asarray(Image.open(io.BytesIO(b)).convert('RGB'))[..., ::-1] for b in blobs]
    except ImportError:
        pil = None
    blobs = blobs + [full]
    dec = sncal.JpegDecoder(540, 960, max_batch=8, threads=3, device=cuda)
    for _ in range(3):                                  # both staging slots, twice
        out = dec.decode(blobs)
    got = out.cpu().numpy()
    assert np.array_equal(got[0], want) and np.array_equal(got[-1], want)
    for i in range(1, len(blobs) - 1):
        assert np.array_equal(got[i], oj.decode_bgr(blobs[i])), i
        if pil is not None:
            assert np.array_equal(got[i], pil[i]), i
    # the decoded batch is what the u8 forward takes (make_submit.py:62-66: imread -> ToTensor -> predict)
    net = sncal.HRNetHeatmap('hrnet_w18', dtype='fp32', device=cuda)
    net.load_state_dict(hr.seeded_state_dict(hr.load_config('hrnet_w18'), 3, 4.0))
    _, k_u8 = net.forward(out[:2], want_heat=False, decode_size=(540, 960))
    x = torch.from_numpy(np.ascontiguousarray(got[:2])).permute(0, 3, 1, 2).contiguous().to(torch.float32).div(255)   # ToTensor, on the host
    _, k_f = net.forward(x.to(cuda), want_heat=False, decode_size=(540, 960))
    assert torch.equal(k_u8, k_f)
    dec.close()


def test_bad_frames_fail_loudly_and_name_the_frame(sncal, cuda, gold_dir):
    g, _ = _cases(gold_dir)
    good = g['jpg.48x64_420_q95_r0'].tobytes()
    dec = sncal.JpegDecoder(48, 64, max_batch=4, device=cuda)
    with pytest.raises(sncal._lib.SncalError, match='frame 1.*29x37|frame 1.*37x29'):
        dec.decode([good, g['jpg.29x37_420_q95_r0'].tobytes()])
    with pytest.raises(sncal._lib.SncalError, match='frame 2.*SOF2'):
        dec.decode([good, good, g['jpg.progressive'].tobytes()])
    with pytest.raises(sncal._lib.SncalError, match='max_batch'):
        dec.decode([good] * 5)
    assert dec.decode([]).shape == (0, 48, 64, 3)
    out = dec.decode([good])                            # still usable after the failures
    assert np.array_equal(out[0].cpu().numpy(), g['bgr.48x64_420_q95_r0'])
    dec.close()


def test_directory_harness_runs_the_make_submit_loop(sncal, cuda, gold_dir, tmp_path):
    """make_submit.py:42-75 counterpart: JPEG files -> JpegDecoder -> predict -> batched solve -> camera_<frame>.json.
    Five frames in batches of two (a ragged last batch); the keypoints the loop solved from are those of the
    reference-shaped path (decoded pixels -> ToTensor -> model.predict), and the files written are exactly the
    cameras CameraCreator returns for them."""
    g, _ = _cases(gold_dir)
    full = g['jpg.full'].tobytes()
    img_dir, save_dir = tmp_path / 'imgs', tmp_path / 'out'
    img_dir.mkdir()
    names = [f'{i:05d}.jpg' for i in range(5)]
    for n in names:
        (img_dir / n).write_bytes(full)
    (img_dir / 'notes.txt').write_text('not a frame')
    cfg = hr.load_config('hrnet_w18')
    ck = {'model_name': 'HRNetMetaModel',
          'params': {'nn_module': {'hrnet_config': cfg, 'num_refinement_stages': 0, 'num_heatmaps': 58},
                     'prediction_transform': {'size': [540, 960]}, 'device': 'cuda:0'},
          'nn_state_dict': hr.seeded_state_dict(cfg, 3, 4.0)}
    path = str(tmp_path / 'model.pth')
    torch.save(ck, path)
    model = sncal.load_model(path, loss=None, optimizer=None, device='cuda:0', dtype='fp32')
    cal = sncal.submit.default_calibrator()
    res = sncal.submit.make_submit(str(img_dir), model, cal, str(save_dir), batch_size=2, decoder_threads=2)
    assert res['frames'] == 5 and 0.0 <= res['completeness'] <= 1.0
    x = torch.from_numpy(np.ascontiguousarray(oj.decode_bgr(full))).permute(2, 0, 1).contiguous().to(torch.float32).div(255)
    pred = model.predict(torch.stack([x] * 5))
    cams = cal.solve_batch(pred.cpu().numpy(), names)
    want = sorted('camera_' + n.replace('.jpg', '.json') for n, c in zip(names, cams) if c is not None)
    assert sorted(os.listdir(save_dir)) == want and res['written'] == len(want)


def test_directory_harness_skips_files_the_decoder_cannot_take(sncal, cuda, gold_dir, tmp_path):
    """A file that is not a JPEG and a frame of another size do not end the run (cv2.imread would not raise either, make_submit.py:62): they are
    skipped with a warning, get no camera file and count against completeness."""
    g, _ = _cases(gold_dir)
    full = g['jpg.full'].tobytes()
    small = g['jpg.48x64_420_q95_r0'].tobytes()
    img_dir, save_dir = tmp_path / 'imgs', tmp_path / 'out'
    img_dir.mkdir()
    (img_dir / '00000.jpg').write_bytes(full)
    (img_dir / '00001.jpg').write_bytes(b'\xff\xd8' + bytes(range(256)) * 4)     # not a JPEG stream behind the SOI marker
    (img_dir / '00002.jpg').write_bytes(small)                          # 64x48 where the run's frames are 960x540
    (img_dir / '00003.jpg').write_bytes(full)
    cfg = hr.load_config('hrnet_w18')
    ck = {'model_name': 'HRNetMetaModel',
          'params': {'nn_module': {'hrnet_config': cfg, 'num_refinement_stages': 0, 'num_heatmaps': 58},
                     'prediction_transform': {'size': [540, 960]}, 'device': 'cuda:0'},
          'nn_state_dict': hr.seeded_state_dict(cfg, 3, 4.0)}
    path = str(tmp_path / 'model.pth')
    torch.save(ck, path)
    model = sncal.load_model(path, loss=None, optimizer=None, device='cuda:0', dtype='fp32')
    names = ['00000.jpg', '00001.jpg', '00002.jpg', '00003.jpg']
    with pytest.warns(UserWarning, match='skipped'):
        res = sncal.submit.make_submit(str(img_dir), model, sncal.submit.default_calibrator(), str(save_dir), batch_size=2,
                                       img_names=names, decoder_threads=2)
    assert res['frames'] == 4 and sorted(res['skipped']) == ['00001.jpg', '00002.jpg']
    assert res['written'] <= 2 and all(f in ('camera_00000.json', 'camera_00003.json') for f in os.listdir(save_dir))


def test_directory_harness_survives_an_odd_first_file_and_a_stream_damaged_behind_its_headers(sncal, cuda, gold_dir, tmp_path):
    """The run's frame size is the modal size of the probeable head of the listing, not the first file's; a stream whose entropy
    data is corrupt passes the probe, fails inside the batch decode, and the per-frame re-decode drops only that frame.  (A stream
    that merely ENDS early is decoded like libjpeg does it -- zero coefficients for the missing MCUs -- and is not an error.)
    A run that lost more than max_skip_fraction of its frames raises."""
    g, _ = _cases(gold_dir)
    full = g['jpg.full'].tobytes()
    small = g['jpg.48x64_420_q95_r0'].tobytes()
    sos = full.find(b'\xff\xda')
    assert sos > 0
    start = sos + 2 + ((full[sos + 2] << 8) | full[sos + 3])             # first byte of the entropy-coded segment
    cut = full[:start + 500] + b'\xff\x00' * 64 + full[start + 628:]     # 1024 one-bits: no Huffman code of the tables is that long
    img_dir, save_dir = tmp_path / 'imgs', tmp_path / 'out'
    img_dir.mkdir()
    names = ['00000.jpg', '00001.jpg', '00002.jpg', '00003.jpg', '00004.jpg']
    for n, blob in zip(names, (small, full, cut, full, full)):            # thumbnail FIRST; the cut stream shares a batch with a good one
        (img_dir / n).write_bytes(blob)
    cfg = hr.load_config('hrnet_w18')
    ck = {'model_name': 'HRNetMetaModel',
          'params': {'nn_module': {'hrnet_config': cfg, 'num_refinement_stages': 0, 'num_heatmaps': 58},
                     'prediction_transform': {'size': [540, 960]}, 'device': 'cuda:0'},
          'nn_state_dict': hr.seeded_state_dict(cfg, 3, 4.0)}
    path = str(tmp_path / 'model.pth')
    torch.save(ck, path)
    model = sncal.load_model(path, loss=None, optimizer=None, device='cuda:0', dtype='fp32')
    assert sncal.submit.run_frame_size(str(img_dir), names) == (540, 960)
    with pytest.warns(UserWarning, match='skipped'):
        res = sncal.submit.make_submit(str(img_dir), model, sncal.submit.default_calibrator(), str(save_dir), batch_size=2,
                                       img_names=names, decoder_threads=2)
    assert res['frames'] == 5 and sorted(res['skipped']) == ['00000.jpg', '00002.jpg']
    assert all(f in ('camera_00001.json', 'camera_00003.json', 'camera_00004.json') for f in os.listdir(save_dir))
    with pytest.warns(UserWarning, match='skipped'), pytest.raises(sncal._lib.SncalError, match='skipped'):
        sncal.submit.make_submit(str(img_dir), model, sncal.submit.default_calibrator(), str(tmp_path / 'out2'), batch_size=2,
                                 img_names=['00000.jpg', '00002.jpg', '00001.jpg'], decoder_threads=2)
