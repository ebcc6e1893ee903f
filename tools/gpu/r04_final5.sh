#!/bin/bash
# round 4, final evidence after the fused env step, call 1: env / bench tests, then rocprofv3 kernel stats + PMC passes of every single-GPU configuration
set -u -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_environment.py tests/test_gpu_full_size.py tests/test_gpu_bench.py -x -q -m gpu 2>&1 | tail -n 2 || exit 1
bash profiles/run_profiles.sh r04 "" 50 > gpurun_out/run_profiles_r04.log 2>&1; echo "r04 $(grep -c 'rc=0' gpurun_out/run_profiles_r04.log)"
bash profiles/run_profiles.sh r04_ref300 "--envs-per-gpu 512 --frame 300 --spp 100" 20 > gpurun_out/run_profiles_r04_ref300.log 2>&1; echo "ref300 $(grep -c 'rc=0' gpurun_out/run_profiles_r04_ref300.log)"
bash profiles/run_profiles.sh r04_c4 "--envs-per-gpu 128 --frame 512 --spp 64" 20 > gpurun_out/run_profiles_r04_c4.log 2>&1; echo "c4 $(grep -c 'rc=0' gpurun_out/run_profiles_r04_c4.log)"
bash profiles/run_profiles.sh r04_c1 "--envs-per-gpu 256 --frame 128 --spp 4" 200 > gpurun_out/run_profiles_r04_c1.log 2>&1; echo "c1 $(grep -c 'rc=0' gpurun_out/run_profiles_r04_c1.log)"
