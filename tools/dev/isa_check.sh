#!/bin/bash
# usage: tools/dev/isa_check.sh <csrc file> -> register counts / spills of every kernel in it (ISA left under /tmp/isa)
mkdir -p /tmp/isa; cd /root/repo/soccernet-calibration-sportlight_amd/csrc
f=$1; b=${f%.*}
hipcc -x hip -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -c $f -o /tmp/isa/$b.o -save-temps=obj 2>&1 | grep -E "error" -A3 | head -30
grep -E "^\s+\.(vgpr_count|vgpr_spill_count|sgpr_spill_count|name):" /tmp/isa/$b-hip-amdgcn-amd-amdhsa-gfx950.s | paste - - - - | sed 's/\s\+/ /g'
