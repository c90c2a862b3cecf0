#!/bin/bash
for v in new prev; do
if [ $v = new ]; then unset SNCAL_LIB_PATH; else export SNCAL_LIB_PATH=tools/ab/libsncal_$v.so; fi
echo "--- $v"; timeout 300 python tools/dev/lm_iter_probe.py 1 2>&1 | tail -3
done
