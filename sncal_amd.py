"""Import alias: ``import sncal_amd`` loads the package directory ``soccernet-calibration-sportlight_amd``
(a hyphen cannot appear in an ``import`` statement)."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'soccernet-calibration-sportlight_amd')
_spec = importlib.util.spec_from_file_location('sncal_amd', os.path.join(_dir, '__init__.py'),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules['sncal_amd'] = _mod
_spec.loader.exec_module(_mod)
