#!/bin/bash
# A/B: the library compiled for gfx950:xnack- against the default (xnack any)
R=$GRAFT_REPO_ROOT; cd $R
rocminfo 2>/dev/null | grep -i "xnack\|gfx950" | head -4
for rep in 1 2; do
  for lib in "" tools/ab/libsncal_xnackoff.so; do
    if [ -z "$lib" ]; then unset SNCAL_LIB_PATH; tag=default; else export SNCAL_LIB_PATH=$R/$lib; tag=xnackoff; fi
    DEV_TOP=5 timeout 120 python tools/dev_bench.py 64 fp16x3 3 2>&1 | grep -v "^W\|amdgpu.ids" | head -6 | sed "s/^/$tag /"
  done
done
