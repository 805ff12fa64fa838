bash tools/gpu_check.sh r02n
