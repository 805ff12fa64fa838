export TMPDIR=/tmp
O=gpurun_out/r03q; mkdir -p $O
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python bench.py --quick --no-cpu-baseline --exec-steps 2 --exec-warmup 1 --steps 5 --warmup 3 > $O/prof_bench.json 2> $O/prof.err; echo "rocprof rc=$?"
find $O/prof -name "*kernel_stats*.csv" | head -1 | xargs -r -I{} cp {} $O/kernel_stats.csv
rm -rf $O/prof
head -n 30 $O/kernel_stats.csv | cut -c1-150
