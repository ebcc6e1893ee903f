#!/bin/bash
# round 4, call 53: where one pixel per thread with cooperative tails overtakes the kernel without them
set -u
run() { local name=$1; shift
  timeout -k 10 300 python bench.py --no-pmc --no-cpu-baseline --no-kernel-timing "$@" | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$name', round(d['value'],1), 'env-steps/s', round(d['ms_per_step']*1000,1), 'us/step', flush=True)"; }
for mode in coop1 nocoop; do
  if [ $mode = nocoop ]; then export REINFOCUS_RENDER_COOP=0; else unset REINFOCUS_RENDER_COOP; fi
  export REINFOCUS_RENDER_SETS=1
  for n in 6 8 12 16; do run ${mode}_e${n}_300 --envs-per-gpu $n --frame 300 --spp 100 --steps 150 --warmup 10; done
  for n in 4 8 16; do run ${mode}_e${n}_256 --envs-per-gpu $n --frame 256 --spp 16 --steps 500 --warmup 10; done
  for n in 8 32; do run ${mode}_e${n}_128 --envs-per-gpu $n --frame 128 --spp 16 --steps 500 --warmup 10; done
done 2>&1 | tee gpurun_out/r04_aq.txt
