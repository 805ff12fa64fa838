#!/bin/bash
OUT=gpurun_out/r04k; mkdir -p $OUT; export TMPDIR=/tmp; ROOT=$PWD
cd /tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $ROOT/$OUT/prof -- python $ROOT/tools/step_only.py 10 0 0 > $ROOT/$OUT/step.log 2>&1; echo rc=$?
f=$(find $ROOT/$OUT/prof -name "*kernel_trace.csv" | head -1)
python $ROOT/tools/trace_positions.py $f "k_conv3x3_fwd<2, 1>" 60 > $ROOT/$OUT/pos_fwd21.txt
python $ROOT/tools/trace_positions.py $f "k_conv3x3s2_dgrad" 20 > $ROOT/$OUT/pos_s2dgrad.txt
python $ROOT/tools/trace_positions.py $f "k_conv3x3_fwd<1, 2>" 20 > $ROOT/$OUT/pos_fwd12.txt
python $ROOT/tools/trace_positions.py $f "k_gemm_nt<2>" 60 > $ROOT/$OUT/pos_gemm_nt.txt
python $ROOT/tools/trace_underfilled.py $f 14 > $ROOT/$OUT/underfilled.txt
python $ROOT/tools/trace_underfilled.py $f 14 0 k_bn > $ROOT/$OUT/bn.txt
head -2 $f | cut -c1-400 > $ROOT/$OUT/trace_head.txt
python $ROOT/tools/trace_biggaps.py $f 1297 8 > $ROOT/$OUT/biggaps.txt
wc -l $f > $ROOT/$OUT/nkern.txt
rm -rf $ROOT/$OUT/prof
cd $ROOT; tail -1 $OUT/step.log; cat $OUT/bn.txt
