#!/bin/bash
# round 4, call 65: the replayed step's io straight from / to the pinned host buffer (no copy nodes)
set -u -o pipefail
timeout -k 10 900 python -m pytest tests/test_gpu_environment.py tests/test_gpu_strategy_cases.py tests/test_gpu_full_size.py -x -q -m gpu 2>&1 | tail -n 3 || exit 1
timeout -k 10 600 python tools/soak_env.py 150 47 2>&1 | tail -n 1 || exit 1
run() { local name=$1 zc=$2; shift 2
  REINFOCUS_ENV_ZEROCOPY=$zc timeout -k 10 300 python bench.py --no-pmc --no-cpu-baseline --no-kernel-timing "$@" | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$name zero-copy=$zc', round(d['value'],1), 'env-steps/s', round(d['ms_per_step']*1000,1), 'us/step', flush=True)"; }
for rep in 1 2; do for zc in 0 1; do
  run c0 $zc --envs-per-gpu 1 --frame 64 --spp 1 --steps 3000 --warmup 20
  run c1 $zc --envs-per-gpu 256 --frame 128 --spp 4 --steps 2000 --warmup 20
  run e1_300 $zc --envs-per-gpu 1 --frame 300 --spp 100 --steps 300 --warmup 10
  run head $zc --steps 12 --warmup 2
done; done 2>&1 | tee gpurun_out/r04_aw.txt
