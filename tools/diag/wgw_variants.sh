#!/bin/bash
# compile and run tools/diag/wgw_clock.hip in variants on the GPU box: what each part of a step of k_wino_wgrad costs
#   usage: tools/diag/wgw_variants.sh ["-DFLAGS of variant 1" "-DFLAGS of variant 2" ...]     (default: the product's configuration,
#   the statements-per-gap sweep, and the same loop reading one row over and over)
cd $(dirname $0)/../..
if [ $# -eq 0 ]; then set -- "" "-DT2O_WGW_VPG=2" "-DT2O_WGW_VPG=8" "-DT2O_WGW_VPG=19" "-DT2O_WGW_SAMEROWS" "-DT2O_WGW_NO_XFORM"; fi
for v in "$@"; do
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -std=c++17 -DT2O_WGW_DIAG $v -Iinclude -o /tmp/wgw_clock tools/diag/wgw_clock.hip 2>/dev/null || { echo "compile failed: $v"; continue; }
  echo "== variant [$v]"
  /tmp/wgw_clock 64 64 320
  /tmp/wgw_clock 128 32 320
done
