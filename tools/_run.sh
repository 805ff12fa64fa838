mkdir -p gpurun_out/r03b; export TMPDIR=/tmp
hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -std=c++17 -Iinclude -o /tmp/wgrad_clock tools/diag/wgrad_clock.hip 2>/dev/null
for a in "128 32" "64 64" "256 16" "512 8"; do timeout 120 /tmp/wgrad_clock $a; done > gpurun_out/r03b/wgrad_clock.txt 2>&1; cat gpurun_out/r03b/wgrad_clock.txt
