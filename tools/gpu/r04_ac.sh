#!/bin/bash
# round 4, call 32: full GPU suite on the single-shape kernel + profile of the single-sphere instance
set -u -o pipefail
timeout -k 10 1000 python -m pytest tests -x -q -m gpu 2>&1 | tail -n 3 || exit 1
for scene in one_rect one_sphere; do
  bash profiles/run_profiles.sh r04_general_$scene "256 256 16 --scene $scene" 0 tools/bench_general.py > gpurun_out/run_profiles_r04_general_$scene.log 2>&1; echo "general $scene $(grep -c 'rc=0' gpurun_out/run_profiles_r04_general_$scene.log)"
done
