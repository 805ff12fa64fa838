#!/bin/bash
# k_wino_fused under different numbers of transform statements per MFMA gap (tools/diag/wf_clock.hip, per-chunk cycle stamps)
#   usage: tools/diag/wf_variants.sh [values of T2O_WF_VPG ...]      (default: 2 4 8 16 32)
cd $(dirname $0)/../..
if [ $# -eq 0 ]; then set -- 2 4 8 16 32; fi
for v in "$@"; do
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -std=c++17 -DT2O_WF_DIAG -DT2O_WF_VPG=$v -Iinclude -o /tmp/wf_clock tools/diag/wf_clock.hip 2>/dev/null || { echo "compile failed: $v"; continue; }
  echo "== T2O_WF_VPG=$v"
  /tmp/wf_clock 64 64
  /tmp/wf_clock 128 32 | head -2
done
