"""MI355X-native per-frame soccer camera calibration hot path (HRNet heatmaps -> decode -> camera solve).

Host mirror of the reference's Python call surfaces over libsncal.so (HIP, gfx950).
Import as ``import sncal_amd`` (alias module at the repo root) -- the directory name carries a hyphen.
"""
from . import _lib  # noqa: F401
from .transforms import HRNetPredictionTransform, EHMPredictionTransform  # noqa: F401
from .hrnet import HRNetHeatmap, load_config  # noqa: F401
from .metamodel import HRNetMetaModel, EHMMetaModel, load_model  # noqa: F401
from .camera import Camera  # noqa: F401
from .prediction import CameraCreator  # noqa: F401
from .pitch import PITCH_POINTS, INTERSECTON_TO_PITCH_POINTS  # noqa: F401
from . import lines, pitch, prediction, camera, synth, dist, pipeline, interop, evaluate, jpeg, submit, loss, annotations  # noqa: F401
from .evaluate import CameraEvaluator  # noqa: F401
from .jpeg import JpegDecoder  # noqa: F401
from .pipeline import CalibrationPipeline  # noqa: F401

__all__ = ['HRNetPredictionTransform', 'EHMPredictionTransform', 'HRNetHeatmap', 'load_config',
           'HRNetMetaModel', 'EHMMetaModel', 'load_model', 'Camera', 'CameraCreator', 'PITCH_POINTS',
           'INTERSECTON_TO_PITCH_POINTS']
