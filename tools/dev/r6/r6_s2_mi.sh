#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
for mi in 0 3; do
  SNCAL_PROFILE_DETAIL=1 SNCAL_FORCE_MI_S2=$mi DEV_TOP=80 timeout 120 python tools/dev_bench.py 64 fp16x3 3 2>&1 | grep "k3,s2\|shared_s2" | sed "s/^/mi=$mi /"
done
