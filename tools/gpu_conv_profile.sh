#!/bin/bash
# One GPU-box visit for the convolution kernels: own vs library timings, in-kernel stamps, the instruction-cost
# microbenchmarks the kernel design rests on, and the matrix-pipe occupancy counters.  Logs -> gpurun_out/<tag>/.
TAG=${1:-conv}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
F="-O3 --offload-arch=gfx950 -std=c++17"
python tools/bench_conv.py 64 fdws 2>&1 | grep -v amdgpu.ids > $OUT/conv_vs_miopen.txt; cat $OUT/conv_vs_miopen.txt
hipcc $F -ffp-contract=off -DT2O_CONV_DIAG -Iinclude -o /tmp/wgrad_clock tools/diag/wgrad_clock.hip 2>/dev/null
hipcc $F -ffp-contract=off -DT2O_CONV_DIAG -Iinclude -o /tmp/fwd_clock tools/diag/fwd_clock.hip 2>/dev/null
hipcc $F -o /tmp/mfma_clock tools/diag/mfma_clock.hip 2>/dev/null
hipcc $F -o /tmp/valu_beside_mfma tools/diag/valu_beside_mfma.hip 2>/dev/null
hipcc $F -o /tmp/glds_in_mfma tools/diag/glds_in_mfma.hip 2>/dev/null
hipcc $F -o /tmp/glds_issue tools/diag/glds_issue.hip 2>/dev/null
{ for a in "64 64" "128 32" "256 16" "512 8"; do timeout 120 /tmp/wgrad_clock $a; done; } > $OUT/wgrad_in_kernel.txt 2>&1
{ for a in "64 64 2" "128 32 2" "256 16 2" "512 8 1"; do timeout 120 /tmp/fwd_clock $a; done; } > $OUT/fwd_in_kernel.txt 2>&1
timeout 120 /tmp/mfma_clock > $OUT/mfma_fp32_bare_loop.txt 2>&1
timeout 120 /tmp/valu_beside_mfma > $OUT/valu_beside_mfma.txt 2>&1
timeout 120 /tmp/glds_in_mfma > $OUT/glds_in_mfma.txt 2>&1
timeout 120 /tmp/glds_issue > $OUT/glds_issue.txt 2>&1
# counters: own process per pass, --pmc only with --kernel-trace
C="SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU"
timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_cal -- /tmp/mfma_clock > /dev/null 2>&1; echo "pmc cal rc=$?"
timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_conv -- python tools/bench_conv.py 64 fdws > /dev/null 2>&1; echo "pmc conv rc=$?"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_conv -- python tools/bench_conv.py 64 fdws > /dev/null 2>&1; echo "stats rc=$?"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_cal -- /tmp/mfma_clock > /dev/null 2>&1
python tools/mfma_busy.py $OUT > $OUT/mfma_busy.txt; cat $OUT/mfma_busy.txt
rm -rf $OUT/pmc_cal $OUT/pmc_conv $OUT/stats_conv $OUT/stats_cal
