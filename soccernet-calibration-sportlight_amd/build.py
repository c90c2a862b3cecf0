"""Build libsncal.so (HIP, gfx950 only) in-tree with hipcc.

    python -m build            (from this directory)   or   __graft_entry__.build()

Every csrc/*.hip / *.cpp is compiled to build/<name>.o (recompiled only when the source or a header is
newer) and linked into <package>/libsncal.so.  hipcc cross-compiles without a GPU.
"""
import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, 'csrc')
OBJ = os.path.join(PKG, 'build')
LIB = os.path.join(PKG, 'libsncal.so')
ARCH = 'gfx950'
CXXFLAGS = ['-O3', '-std=c++17', '-fPIC', f'--offload-arch={ARCH}', '-Wall', '-Wno-unused-function',
            '-ffp-contract=off']


def _hipcc():
    for c in (shutil.which('hipcc'), '/opt/rocm/bin/hipcc'):
        if c and os.path.exists(c):
            return c
    raise RuntimeError('hipcc not found (ROCm toolchain required to build libsncal.so)')


def _newer(a, b):
    return not os.path.exists(b) or os.path.getmtime(a) > os.path.getmtime(b)


def build(verbose=False, force=False, jobs=None, x3_f16=None, lib=None, obj=None, extra_flags=()):
    """x3_f16=0 / 1 builds the split-arithmetic engine with bf16 / fp16 splits (default: the source's default, fp16) -- an A/B build
    goes to its own `lib` path and `obj` directory (tools/ab_build.py)."""
    global OBJ, LIB
    OBJ, LIB = obj or OBJ, lib or LIB
    hipcc = _hipcc()
    flags = list(CXXFLAGS) + ([f'-DSNCAL_X3_F16={int(x3_f16)}'] if x3_f16 is not None else []) + list(extra_flags)
    os.makedirs(OBJ, exist_ok=True)
    srcs = sorted(f for f in os.listdir(CSRC) if f.endswith(('.hip', '.cpp')))
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(('.hpp', '.h', '.inc'))]
    hdrs.append(os.path.join(os.path.dirname(PKG), 'include', 'sncal.h'))
    hdr_time = max(os.path.getmtime(h) for h in hdrs)
    procs, objs = [], []
    jobs = jobs or min(6, os.cpu_count() or 1)
    for s in srcs:
        src = os.path.join(CSRC, s)
        o = os.path.join(OBJ, s.rsplit('.', 1)[0] + '.o')
        objs.append(o)
        if force or _newer(src, o) or os.path.getmtime(o) < hdr_time:
            cmd = [hipcc, '-x', 'hip', *flags, '-c', src, '-o', o]
            if verbose:
                print(' '.join(cmd), flush=True)
            procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
            while len([p for _, p in procs if p.poll() is None]) >= jobs:
                [p.wait() for _, p in procs if p.poll() is None][:1]
    failed = False
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0 or (verbose and out):
            sys.stderr.write(out.decode(errors='replace'))
        if p.returncode != 0:
            failed = True
            sys.stderr.write(f'hipcc failed on {s}\n')
    if failed:
        raise RuntimeError('libsncal.so build failed')
    if procs or not os.path.exists(LIB):
        cmd = [hipcc, '-shared', '-fPIC', f'--offload-arch={ARCH}', *objs, '-o', LIB]
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == '__main__':
    print(build(verbose=True, force='--force' in sys.argv))
