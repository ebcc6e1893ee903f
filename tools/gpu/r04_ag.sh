#!/bin/bash
# round 4, call 40: strip tiles for widths that are not multiples of 64 (render_kernel_coop2_strip)
set -u -o pipefail
timeout -k 10 1000 python -m pytest tests/test_gpu_parity.py tests/test_gpu_environment.py tests/test_gpu_full_size.py tests/test_gpu_bench.py tests/test_gpu_notebook_figures.py -x -q -m gpu 2>&1 | tail -n 5 || exit 1
run() { local name=$1 strip=$2; shift 2
  REINFOCUS_RENDER_STRIP=$strip timeout -k 10 300 python bench.py --no-pmc --no-cpu-baseline "$@" | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$name strip=$strip', round(d['value'],1), round(d['roofline']['samples_per_s']/1e9,2), d['roofline']['kernel'], flush=True)"; }
for rep in 1 2 3; do for strip in 0 1; do
  run ref300 $strip --envs-per-gpu 512 --frame 300 --spp 100 --steps 8 --warmup 2
done; done
for rep in 1 2; do for strip in 0 1; do
  run f100 $strip --envs-per-gpu 2048 --frame 100 --spp 16 --steps 20 --warmup 2
  run f600 $strip --envs-per-gpu 256 --frame 600 --spp 16 --steps 10 --warmup 2
  run f200 $strip --envs-per-gpu 1024 --frame 200 --spp 16 --steps 10 --warmup 2
  run f400 $strip --envs-per-gpu 512 --frame 400 --spp 16 --steps 10 --warmup 2
done; done 2>&1 | tee gpurun_out/r04_ag.txt
