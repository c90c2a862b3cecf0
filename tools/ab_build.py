#!/usr/bin/env python3
"""Second build of libsncal.so with the OTHER 16-bit split type, for A/B runs (SNCAL_LIB_PATH=tools/ab/libsncal_<name>.so):
    python tools/ab_build.py bf16x3 | fp16x3"""
import importlib.util, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location('sncal_build', os.path.join(ROOT, 'soccernet-calibration-sportlight_amd', 'build.py'))
mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
name = sys.argv[1] if len(sys.argv) > 1 else 'bf16x3'
os.makedirs(os.path.join(ROOT, 'tools', 'ab'), exist_ok=True)
# python tools/ab_build.py <name> -DFOO=1 ...: any other name = the default split type with the given extra compiler flags
extra = [a for a in sys.argv[2:] if a.startswith('-')]
print(mod.build(verbose=False, x3_f16=(name == 'fp16x3') if name in ('fp16x3', 'bf16x3') else None, lib=os.path.join(ROOT, 'tools', 'ab', f'libsncal_{name}.so'),
                obj=os.path.join(ROOT, 'soccernet-calibration-sportlight_amd', 'build', name), extra_flags=extra))
