bash tools/gpu_check.sh r03i
bash tools/gpu_pmc.sh r03i_pmc
