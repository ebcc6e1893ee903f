#!/bin/bash
# round 4, call 49: where three pixels per thread overtake one (blocks of the launch)
set -u
run() { local name=$1 sets=$2; shift 2
  if [ $sets = 1 ]; then export REINFOCUS_RENDER_SETS=1; else unset REINFOCUS_RENDER_SETS; fi
  timeout -k 10 300 python bench.py --no-pmc --no-cpu-baseline --no-kernel-timing "$@" | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$name sets=$sets', round(d['value'],1), 'env-steps/s', round(d['ms_per_step']*1000,1), 'us/step', flush=True)"; }
for n in 6 8 10 12; do for sets in 3 1; do run e${n}_300 $sets --envs-per-gpu $n --frame 300 --spp 100 --steps 200 --warmup 10; done; done 2>&1 | tee gpurun_out/r04_am.txt
for n in 4 8 12; do for sets in 3 1; do run e${n}_256 $sets --envs-per-gpu $n --frame 256 --spp 16 --steps 500 --warmup 10; done; done 2>&1 | tee -a gpurun_out/r04_am.txt
for n in 2 4; do for sets in 3 1; do run e${n}_512 $sets --envs-per-gpu $n --frame 512 --spp 64 --steps 100 --warmup 5; done; done 2>&1 | tee -a gpurun_out/r04_am.txt
for n in 16 32 64; do for sets in 3 1; do run e${n}_128 $sets --envs-per-gpu $n --frame 128 --spp 16 --steps 500 --warmup 10; done; done 2>&1 | tee -a gpurun_out/r04_am.txt
