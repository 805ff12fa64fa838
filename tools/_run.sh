F="-O3 --offload-arch=gfx950 -ffp-contract=off -std=c++17 -DT2O_CONV_DIAG -Iinclude"
hipcc $F -o /tmp/wc0 tools/diag/wgrad_clock.hip 2>&1 | grep -i " error"
for p in 1 3; do hipcc $F -DT2O_WGRAD_PRIO=$p -o /tmp/wc$p tools/diag/wgrad_clock.hip 2>&1 | grep -i " error"; done
for p in 0 1 3; do echo "== prio $p"; for a in "64 64" "128 32" "512 8"; do timeout 120 /tmp/wc$p $a | grep "kernel\|dispatch order"; done; done
