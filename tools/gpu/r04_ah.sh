#!/bin/bash
# round 4, call 41: strip kernel, main tiles 64 x 12 against 128 x 6 (where the main width allows)
set -u
run() { local name=$1 wide=$2; shift 2
  if [ $wide = 1 ]; then export RF_XP_STRIP_WIDE=1; else unset RF_XP_STRIP_WIDE; fi
  timeout -k 10 300 python bench.py --no-pmc --no-cpu-baseline "$@" | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$name wide=$wide', round(d['value'],1), round(d['roofline']['samples_per_s']/1e9,2), d['roofline']['kernel'], flush=True)"; }
for rep in 1 2 3; do for wide in 0 1; do
  run ref300 $wide --envs-per-gpu 512 --frame 300 --spp 100 --steps 8 --warmup 2
  run f400 $wide --envs-per-gpu 512 --frame 400 --spp 16 --steps 10 --warmup 2
done; done 2>&1 | tee gpurun_out/r04_ah.txt
