#!/bin/bash
# round 3, GPU call J: general renderer profiles (kernel trace + PMC) per scene
set -u
for scene in one_rect one_sphere mixed; do
  bash profiles/run_profiles.sh r03_general_$scene "256 256 16 --scene $scene" 0 tools/bench_general.py > gpurun_out/run_profiles_r03_general_$scene.log 2>&1; echo "general $scene done: $(grep -c 'rc=0' gpurun_out/run_profiles_r03_general_$scene.log) passes ok"
done
