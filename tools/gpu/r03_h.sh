#!/bin/bash
# round 3, GPU call H: committed evidence -- kernel-trace stats + PMC passes per configuration
set -u
bash profiles/run_profiles.sh r03 "" 50 > gpurun_out/run_profiles_r03.log 2>&1; tail -9 gpurun_out/run_profiles_r03.log | head -8
bash profiles/run_profiles.sh r03_ref300 "--envs-per-gpu 512 --frame 300 --spp 100" 20 > gpurun_out/run_profiles_r03_ref300.log 2>&1; echo "ref300 done"
bash profiles/run_profiles.sh r03_c4 "--envs-per-gpu 128 --frame 512 --spp 64" 20 > gpurun_out/run_profiles_r03_c4.log 2>&1; echo "c4 done"
bash profiles/run_profiles.sh r03_c1 "--envs-per-gpu 256 --frame 128 --spp 4" 200 > gpurun_out/run_profiles_r03_c1.log 2>&1; echo "c1 done"
