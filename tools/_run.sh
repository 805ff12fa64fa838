bash tools/gpu_check.sh r03m
bash tools/gpu_pmc.sh r03m_pmc
