mkdir -p gpurun_out/r02m; export TMPDIR=/tmp; O=gpurun_out/r02m
run() { tag=$1; shift; timeout 600 python -m pytest tests/test_gpu_actor.py -q -x --tb=line -k graphed_encoder "$@" > $O/$tag.log 2>&1; echo "$tag rc=$? $(grep -c 'e+2' $O/$tag.log)"; }
run plain
run nocapture -s
run nowarnings -p no:warnings
run nofault -p no:faulthandler
run noplugins -p no:hypothesispytest -p no:xdist -p no:timeout -p no:cacheprovider
run noassert --assert=plain
