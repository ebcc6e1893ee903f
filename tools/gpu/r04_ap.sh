#!/bin/bash
# round 4, call 52: very small launches: the kernel without cooperative tails (in-wave loops, no barriers) against one pixel per thread with them
set -u
run() { local name=$1; shift
  timeout -k 10 300 python bench.py --no-pmc --no-cpu-baseline --no-kernel-timing "$@" | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$name', round(d['value'],1), 'env-steps/s', round(d['ms_per_step']*1000,1), 'us/step', flush=True)"; }
for mode in coop1 nocoop; do
  if [ $mode = nocoop ]; then export REINFOCUS_RENDER_COOP=0; else unset REINFOCUS_RENDER_COOP; fi
  run ${mode}_e1_300 --envs-per-gpu 1 --frame 300 --spp 100 --steps 300 --warmup 10
  run ${mode}_e2_300 --envs-per-gpu 2 --frame 300 --spp 100 --steps 300 --warmup 10
  run ${mode}_e4_300 --envs-per-gpu 4 --frame 300 --spp 100 --steps 300 --warmup 10
  run ${mode}_e1_256 --envs-per-gpu 1 --frame 256 --spp 16 --steps 500 --warmup 10
  run ${mode}_e1_128 --envs-per-gpu 1 --frame 128 --spp 16 --steps 500 --warmup 10
done 2>&1 | tee gpurun_out/r04_ap.txt
