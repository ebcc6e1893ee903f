#!/bin/bash
# round 4, call 48: few environments (the reference's single-env default: 1 x 300^2 x 100): three pixels per thread against one
set -u
run() { local name=$1 sets=$2; shift 2
  if [ $sets = 1 ]; then export REINFOCUS_RENDER_SETS=1; else unset REINFOCUS_RENDER_SETS; fi
  timeout -k 10 300 python bench.py --no-pmc --no-cpu-baseline --no-kernel-timing "$@" | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$name sets=$sets', round(d['value'],1), 'env-steps/s', round(d['ms_per_step']*1000,1), 'us/step', flush=True)"; }
for sets in 3 1; do
  run e1_300 $sets --envs-per-gpu 1 --frame 300 --spp 100 --steps 300 --warmup 10
  run e4_300 $sets --envs-per-gpu 4 --frame 300 --spp 100 --steps 300 --warmup 10
  run e16_300 $sets --envs-per-gpu 16 --frame 300 --spp 100 --steps 200 --warmup 10
  run e64_300 $sets --envs-per-gpu 64 --frame 300 --spp 100 --steps 100 --warmup 5
  run e1_256 $sets --envs-per-gpu 1 --frame 256 --spp 16 --steps 500 --warmup 10
  run e16_256 $sets --envs-per-gpu 16 --frame 256 --spp 16 --steps 500 --warmup 10
  run e64_256 $sets --envs-per-gpu 64 --frame 256 --spp 16 --steps 300 --warmup 10
  run c1 $sets --envs-per-gpu 256 --frame 128 --spp 4 --steps 1000 --warmup 10
  run c0 $sets --envs-per-gpu 1 --frame 64 --spp 1 --steps 2000 --warmup 10
done 2>&1 | tee gpurun_out/r04_al.txt
