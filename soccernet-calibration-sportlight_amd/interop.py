"""File-level interop with artefacts the reference writes and reads either side of the hot path.

camera_<frame>.json   written by PredictionSaver, /root/reference/src/utils/make_submit.py:28-39, 63-65
                      (json.dump(cam.to_json_parameters(), f, indent=4)); read by
                      /root/reference/baseline/evaluate_camera.py:271-283 and Camera.from_json_parameters
lines pickle          written by /root/reference/src/utils/export_line_result.py:188, 200-201
                      ({image name: {'lines': ..., 'points': ...}}); read by CameraCreator.__init__,
                      /root/reference/src/models/hrnet/prediction.py:105-124, which indexes ['lines'][0]
Host-only helpers (python floats, json, pickle): nothing here touches the GPU.
"""
import json
import os
import pickle
from typing import Dict, Iterable, List, Optional

from .camera import Camera


def camera_json_path(save_dir: str, img_name: str) -> str:
    """make_submit.py:63-65: 'camera_' + image file name with .jpg -> .json."""
    return os.path.join(save_dir, 'camera_' + img_name.replace('.jpg', '.json'))


def save_camera_json(cam: Camera, json_path: str) -> None:
    """Byte-for-byte what PredictionSaver writes (make_submit.py:36-37)."""
    with open(json_path, 'w') as f:
        json.dump(cam.to_json_parameters(), f, indent=4)


def save_cameras(cams: Iterable[Optional[Camera]], img_names: List[str], save_dir: str) -> int:
    """One file per frame that has a camera; returns the number written (the harness's completeness numerator)."""
    os.makedirs(save_dir, exist_ok=True)
    n = 0
    for cam, name in zip(cams, img_names):
        if cam is not None:
            save_camera_json(cam, camera_json_path(save_dir, name))
            n += 1
    return n


def load_camera_json(json_path: str, width: int = 960, height: int = 540) -> Camera:
    """evaluate_camera.py:27-31: Camera(width, height).from_json_parameters(json)."""
    with open(json_path) as f:
        d = json.load(f)
    cam = Camera(width, height)
    cam.from_json_parameters(d)
    return cam


def save_lines_pickle(per_image: Dict[str, dict], path: str, as_list: bool = True) -> None:
    """per_image: {image name: {'lines': {line name: (k, b)}, 'points': {line name: [(x, y, p)]}}}.
    as_list=True wraps both values in one-element lists, the layout CameraCreator actually indexes
    (prediction.py:111 reads ['lines'][0]); as_list=False is export_line_result.py:188's literal layout."""
    out = {}
    for name, rec in per_image.items():
        lines, points = dict(rec['lines']), dict(rec.get('points', {}))
        out[name] = {'lines': [lines], 'points': [points]} if as_list else {'lines': lines, 'points': points}
    with open(path, 'wb') as f:
        pickle.dump(out, f)


def load_lines_pickle(path: str) -> Dict[str, dict]:
    """Either layout -> {image name: {'lines': {...}, 'points': {...}}}."""
    with open(path, 'rb') as f:
        raw = pickle.load(f)
    out = {}
    for name, rec in raw.items():
        lines, points = rec['lines'], rec.get('points', {})
        if isinstance(lines, (list, tuple)):
            lines = lines[0]
        if isinstance(points, (list, tuple)):
            points = points[0] if points else {}
        out[name] = {'lines': lines, 'points': points}
    return out
