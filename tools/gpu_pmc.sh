#!/bin/bash
# PMC passes over bench.py's cfg2 executor leg (fused + materialised kernels): SQ instruction mix / waits, then the
# HBM traffic counters in separate passes (FETCH_SIZE and WRITE_SIZE do not fit one pass).  No trace domains
# beside --kernel-trace are combined with --pmc.
TAG=${1:-pmc}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
CMD="python bench.py --quick --exec-steps 4 --exec-warmup 2 --no-cpu-baseline --no-train"   # executor kernels only
# (no `rocprofv3 -L` here: the launcher re-execs a python helper after the GPU is up, which the pool refuses)
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $OUT/sq -- $CMD > $OUT/sq.log 2>&1; echo "sq rc=$?"
timeout 600 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/sq2 -- $CMD > $OUT/sq2.log 2>&1; echo "sq2 rc=$?"
timeout 600 rocprofv3 --pmc SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_INSTS_VALU_TRANS SQ_INSTS_BRANCH SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_IFETCH SQ_WAIT_IFETCH --kernel-trace --output-format csv -d $OUT/sq3 -- $CMD > $OUT/sq3.log 2>&1; echo "sq3 rc=$?"
# matrix-core utilisation: cycles the MFMA pipe is busy per CU-busy cycle, fp32 MFMA operations (a pass of its own)
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_CYCLES --kernel-trace --output-format csv -d $OUT/mfma -- $CMD > $OUT/mfma.log 2>&1; echo "mfma rc=$?"
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- $CMD > $OUT/fetch.log 2>&1; echo "fetch rc=$?"
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- $CMD > $OUT/write.log 2>&1; echo "write rc=$?"
python tools/pmc_summary.py $OUT > $OUT/summary.txt
# raw per-dispatch CSVs are tens of MB: keep only the per-kernel summary (gpurun merges <= 64 MiB)
rm -rf $OUT/sq $OUT/sq2 $OUT/sq3 $OUT/mfma $OUT/fetch $OUT/write
tail -n 3 $OUT/summary.txt; tail -3 $OUT/sq3.log
