#!/bin/bash
# k_wino_fused in build variants (tools/diag/wf_clock.hip, per-chunk cycle stamps)
#   usage: tools/diag/wf_variants.sh ["-DFLAGS of variant 1" ...]   (default: transform statements per gap, the x pieces' pair)
cd $(dirname $0)/../..
if [ $# -eq 0 ]; then set -- "-DT2O_WF_VPG=2" "-DT2O_WF_VPG=16" "-DT2O_WF_XPAIR=4" "-DT2O_WF_XPAIR=2" "-DT2O_WF_XPAIR=1"; fi
for v in "$@"; do
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -std=c++17 -DT2O_WF_DIAG $v -Iinclude -o /tmp/wf_clock tools/diag/wf_clock.hip 2>/dev/null || { echo "compile failed: $v"; continue; }
  echo "== $v"
  /tmp/wf_clock 64 64
  /tmp/wf_clock 128 32 | head -2
  /tmp/wf_clock 256 16 | head -2
done
