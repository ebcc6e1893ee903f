#!/bin/bash
# round 4, call 59: low-sample configurations (configs[0], configs[1]) on the kernel without cooperative tails
set -u
run() { local name=$1; shift
  timeout -k 10 300 python bench.py --no-pmc --no-cpu-baseline --no-kernel-timing "$@" | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$name', round(d['value'],1), 'env-steps/s', round(d['ms_per_step']*1000,1), 'us/step', flush=True)"; }
for mode in default nocoop; do
  if [ $mode = nocoop ]; then export REINFOCUS_RENDER_COOP=0; else unset REINFOCUS_RENDER_COOP; fi
  run ${mode}_c1 --envs-per-gpu 256 --frame 128 --spp 4 --steps 1000 --warmup 10
  run ${mode}_c0 --envs-per-gpu 1 --frame 64 --spp 1 --steps 2000 --warmup 10
  run ${mode}_64x128x4 --envs-per-gpu 64 --frame 128 --spp 4 --steps 1000 --warmup 10
  run ${mode}_16x256x4 --envs-per-gpu 16 --frame 256 --spp 4 --steps 1000 --warmup 10
  run ${mode}_1x300x4 --envs-per-gpu 1 --frame 300 --spp 4 --steps 1000 --warmup 10
done 2>&1 | tee gpurun_out/r04_av.txt
