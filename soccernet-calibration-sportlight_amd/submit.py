"""Harness counterpart of /root/reference/src/utils/make_submit.py:42-75: image directory ->
`camera_<frame>.json` files + completeness, with every stage of the loop on the GPU path.

    reference loop (per batch of 8)                      here (per batch, default 64)
    cv2.imread + ToTensor + torch.stack   :62-63          JpegDecoder.decode (host Huffman, device IDCT/colour) -> uint8 BGR
    model.predict(tensor).cpu().numpy()   :64             CalibrationPipeline.submit: forward + decode on the main stream,
    16-process pool, CameraCreator per row :65-69          one batched solve on a side stream (overlaps the next batch)
    json.dump(cam.to_json_parameters())   :36-37          interop.save_cameras, written while the next batch runs

    python -m sncal_amd.submit --img-dir DIR --model model.pth --save-dir OUT [--lines-file lines.pkl]
"""
import argparse
import warnings
import os
from typing import List, Optional

os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')      # see pipeline.py: streams must not alias on a hardware queue

import torch  # noqa: E402

from . import _lib
from .interop import save_cameras
from .jpeg import JpegDecoder, probe
from .metamodel import load_model
from .pipeline import CalibrationPipeline
from .pitch import PITCH_POINTS
from .prediction import CameraCreator, camera_from_record


def default_calibrator(lines_file: Optional[str] = None) -> CameraCreator:
    """The CameraCreator make_submit.py:45-50 builds."""
    return CameraCreator(PITCH_POINTS, conf_thresh=0.5, conf_threshs=[0.5, 0.35, 0.2], algorithm='iterative_voter',
                         lines_file=lines_file, max_rmse=55.0, max_rmse_rel=5.0, min_points=5, min_focal_length=10.0,
                         min_points_per_plane=6, min_points_for_refinement=6, reliable_thresh=57)


def run_frame_size(img_dir: str, img_names: List[str], sample: int = 32):
    """(H, W) of the run: the most frequent size among the first `sample` files whose headers the decoder accepts.  One odd file at
    the head of os.listdir's order (a thumbnail, a progressive or damaged stream) therefore neither ends the run nor makes every
    other frame a size mismatch.  Raises when no file of the sample can be probed."""
    sizes = {}
    for n in img_names[:sample]:
        try:
            with open(os.path.join(img_dir, n), 'rb') as f:
                info = probe(f.read())
        except (_lib.SncalError, OSError):
            continue
        key = (info['height'], info['width'])
        sizes[key] = sizes.get(key, 0) + 1
    if not sizes:
        raise _lib.SncalError(f'{img_dir}: none of the first {min(sample, len(img_names))} files is a JPEG the device decoder accepts '
                              '(baseline sequential Huffman, no EXIF rotation)')
    return max(sizes.items(), key=lambda kv: kv[1])[0]


def make_submit(img_dir: str, model, calibrator: CameraCreator, save_dir: str, batch_size: int = 64,
                decoder_threads: int = 0, img_names: Optional[List[str]] = None, max_skip_fraction: float = 0.5) -> dict:
    """The directory loop on a non-blocking stream of its own (JPEG decode, network, result copies), never on the legacy null
    stream: the condition for the pipeline's CU-masked solve streams (pipeline.py).  See _make_submit for the loop itself."""
    own = torch.cuda.Stream(device=model.nn_module.device)
    with torch.cuda.stream(own):
        res = _make_submit(img_dir, model, calibrator, save_dir, batch_size, decoder_threads, img_names, max_skip_fraction)
    own.synchronize()
    return res


def _make_submit(img_dir: str, model, calibrator: CameraCreator, save_dir: str, batch_size: int = 64,
                 decoder_threads: int = 0, img_names: Optional[List[str]] = None, max_skip_fraction: float = 0.5) -> dict:
    """Returns {'frames', 'written', 'completeness', 'skipped'} (make_submit.py:72 prints the completeness).  Files the device
    decoder cannot take are skipped with a warning; when more than `max_skip_fraction` of the run was skipped the function
    raises instead of returning a near-empty result that looks like a bad model."""
    os.makedirs(save_dir, exist_ok=True)
    if img_names is None:
        img_names = [n for n in os.listdir(img_dir) if n.endswith('.jpg')]          # make_submit.py:56
    total = len(img_names)
    if total == 0:
        return {'frames': 0, 'written': 0, 'completeness': 0.0, 'skipped': []}
    H, W = run_frame_size(img_dir, img_names)
    net = model.nn_module
    pt = model.prediction_transform
    pipe = CalibrationPipeline(net, calibrator, decode_size=(pt.H, pt.W))
    dec = JpegDecoder(H, W, max_batch=batch_size, threads=decoder_threads, device=net.device)
    frames = [torch.empty((batch_size, H, W, 3), dtype=torch.uint8, device=net.device) for _ in range(2)]
    pending = []                                                                    # (names, records, event)
    written = 0
    skipped = []                                                                    # frames the device decoder cannot take

    def drain(keep: int):
        nonlocal written
        while len(pending) > keep:
            names, rec, ev = pending.pop(0)
            ev.synchronize()
            cams = [camera_from_record(r, calibrator.img_size) for r in calibrator.records(rec)]
            written += save_cameras(cams, names, save_dir)

    for k, i in enumerate(range(0, total, batch_size)):
        names, blobs = [], []
        for n in img_names[i:i + batch_size]:
            with open(os.path.join(img_dir, n), 'rb') as f:
                blob = f.read()
            # a file the device decoder does not handle (progressive, arithmetic-coded, CMYK, another size than the first frame's, damaged)
            # is skipped with a warning and simply gets no camera file -- what the reference's loop does for a frame without a
            # camera -- instead of ending the run (cv2.imread would decode some of these: decode them elsewhere and upload uint8 frames)
            try:
                fi = probe(blob)
                if (fi['height'], fi['width']) != (H, W):
                    raise _lib.SncalError(f"{fi['width']}x{fi['height']} where the run's frames are {W}x{H}")
            except _lib.SncalError as e:
                warnings.warn(f'{n}: skipped ({e})')
                skipped.append(n)
                continue
            names.append(n)
            blobs.append(blob)
        if not blobs:
            continue
        try:
            x = dec.decode(blobs, frames[k & 1][:len(blobs)])
        except _lib.SncalError:                     # a stream that is damaged behind its headers: find it, drop it, decode the rest
            keep = []
            for n, blob in zip(names, blobs):
                try:
                    dec.decode([blob], frames[k & 1][:1])
                    keep.append((n, blob))
                except _lib.SncalError as e:
                    warnings.warn(f'{n}: skipped ({e})')
                    skipped.append(n)
            if not keep:
                continue
            names, blobs = [n for n, _ in keep], [b for _, b in keep]
            x = dec.decode(blobs, frames[k & 1][:len(blobs)])
        out = pipe.submit(x, names=names)
        pending.append((names, out[1], pipe.last_done))
        drain(1)                                                                    # write batch k-1 while batch k runs
    drain(0)
    pipe.join()
    pipe.check_range()                                                              # the split-fp16 engine's range flag: raises if any batch left the fp16 range
    dec.close()
    if len(skipped) > max_skip_fraction * total:
        raise _lib.SncalError(f'{len(skipped)} of {total} frames were skipped (first: {skipped[0]}): more than the allowed fraction '
                              f'{max_skip_fraction:g}; decode these frames elsewhere and feed uint8 tensors to the pipeline')
    return {'frames': total, 'written': written, 'completeness': written / total, 'skipped': skipped}


def main(argv=None):
    ap = argparse.ArgumentParser(description='Camera calibration of a directory of frames (make_submit.py counterpart)')
    ap.add_argument('--img-dir', required=True)
    ap.add_argument('--model', required=True, help='argus checkpoint of the keypoint model (model_name/params/nn_state_dict)')
    ap.add_argument('--save-dir', required=True)
    ap.add_argument('--lines-file', default=None, help='lines pickle of export_line_result.py (optional)')
    ap.add_argument('--batch-size', type=int, default=64)
    ap.add_argument('--device', default='cuda:0')
    ap.add_argument('--dtype', default=None, choices=['fp16x3', 'bf16x3', 'fp32', 'bf16', 'fp8'],
                    help='default: the fp32-class engine of the build (fp16x3: split-fp16 products, fp32 accumulation -- the exact engine\'s '
                         'keypoint indices on every measured frame); fp32: the reference\'s own arithmetic on the fp32 MFMA, 2.8x slower; '
                         'bf16 / fp8: throughput modes, keypoints may move by one cell on near-ties')
    a = ap.parse_args(argv)
    if not torch.cuda.is_available():
        raise _lib.SncalError('no GPU visible: this package has no CPU path')
    model = load_model(a.model, loss=None, optimizer=None, device=a.device, dtype=a.dtype)
    res = make_submit(a.img_dir, model, default_calibrator(a.lines_file), a.save_dir, batch_size=a.batch_size)
    print(f"Completeness: {res['completeness']:.2f}")
    return res


if __name__ == '__main__':
    main()
