#!/bin/bash
# round 4, call 21: the env step without memset / finalize launches (the environment kernels zero the sums and take the variance from them)
set -u
timeout -k 10 900 python -m pytest tests/test_gpu_environment.py tests/test_gpu_strategy_cases.py tests/test_gpu_full_size.py tests/test_gpu_bench.py -x -q -m gpu 2>&1 | tail -n 2
timeout -k 10 600 python tools/soak_env.py 150 31 2>&1 | tail -n 1
for rep in 1 2 3; do for lib in tools/lib_old.so tools/lib_new.so tools/lib_new2.so; do
  for cfg in "--envs-per-gpu 256 --frame 128 --spp 4 --steps 2000" "--envs-per-gpu 1 --frame 64 --spp 1 --steps 3000"; do
    REINFOCUS_HIP_LIB=$PWD/$lib timeout -k 10 200 python bench.py --no-cpu-baseline --no-pmc --no-kernel-timing --warmup 20 $cfg | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$lib', d['config']['envs_per_gpu'], d['config']['frame'], round(d['value']), round(d['ms_per_step']*1000,1), 'us', flush=True)" || exit 1
  done; done; done 2>&1 | tee gpurun_out/r04_u.txt
REPS=2 bash tools/ab.sh tools/lib_old.so tools/lib_new2.so 2>&1 | tee gpurun_out/r04_u_head.txt
