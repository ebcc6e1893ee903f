#!/bin/bash
# round 4, call 50: launches of few blocks with one pixel per thread, chosen by the library
set -u -o pipefail
timeout -k 10 1000 python -m pytest tests -x -q -m gpu 2>&1 | tail -n 3 || exit 1
run() { local name=$1; shift
  timeout -k 10 300 python bench.py --no-pmc --no-cpu-baseline --no-kernel-timing "$@" | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$name', round(d['value'],1), 'env-steps/s', round(d['ms_per_step']*1000,1), 'us/step', flush=True)"; }
run e1_300 --envs-per-gpu 1 --frame 300 --spp 100 --steps 300 --warmup 10
run e6_300 --envs-per-gpu 6 --frame 300 --spp 100 --steps 200 --warmup 10
run e8_300 --envs-per-gpu 8 --frame 300 --spp 100 --steps 200 --warmup 10
run e1_256 --envs-per-gpu 1 --frame 256 --spp 16 --steps 500 --warmup 10
run c0 --envs-per-gpu 1 --frame 64 --spp 1 --steps 2000 --warmup 10
run c1 --envs-per-gpu 256 --frame 128 --spp 4 --steps 1000 --warmup 10
run head --steps 12 --warmup 2
