#!/bin/bash
# round 4, final evidence on the final tree, call 1: rocprofv3 kernel stats + PMC passes of every single-GPU configuration and of the single-shape general kernels
set -u
mkdir -p gpurun_out
bash profiles/run_profiles.sh r04 "" 50 > gpurun_out/run_profiles_r04.log 2>&1; echo "r04 $(grep -c 'rc=0' gpurun_out/run_profiles_r04.log)"
bash profiles/run_profiles.sh r04_ref300 "--envs-per-gpu 512 --frame 300 --spp 100" 20 > gpurun_out/run_profiles_r04_ref300.log 2>&1; echo "ref300 $(grep -c 'rc=0' gpurun_out/run_profiles_r04_ref300.log)"
bash profiles/run_profiles.sh r04_c4 "--envs-per-gpu 128 --frame 512 --spp 64" 20 > gpurun_out/run_profiles_r04_c4.log 2>&1; echo "c4 $(grep -c 'rc=0' gpurun_out/run_profiles_r04_c4.log)"
bash profiles/run_profiles.sh r04_c1 "--envs-per-gpu 256 --frame 128 --spp 4" 200 > gpurun_out/run_profiles_r04_c1.log 2>&1; echo "c1 $(grep -c 'rc=0' gpurun_out/run_profiles_r04_c1.log)"
for scene in one_rect one_sphere mixed; do
  bash profiles/run_profiles.sh r04_general_$scene "256 256 16 --scene $scene" 0 tools/bench_general.py > gpurun_out/run_profiles_r04_general_$scene.log 2>&1; echo "general $scene $(grep -c 'rc=0' gpurun_out/run_profiles_r04_general_$scene.log)"
done
