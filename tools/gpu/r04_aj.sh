#!/bin/bash
# round 4, call 43: headline / c4 / c1 after the tile-function refactor; profile of the 300 px configuration on the strip kernel
set -u
for rep in 1 2 3; do
timeout -k 10 300 python bench.py --no-pmc --no-cpu-baseline --steps 12 --warmup 2 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('head', round(d['value'],1), round(d['roofline']['samples_per_s']/1e9,2), d['roofline']['kernel'], flush=True)"
done
timeout -k 10 300 python bench.py --no-pmc --no-cpu-baseline --envs-per-gpu 128 --frame 512 --spp 64 --steps 20 --warmup 3 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('c4', round(d['value'],1), round(d['roofline']['samples_per_s']/1e9,2), d['roofline']['kernel'], flush=True)"
bash profiles/run_profiles.sh r04_ref300 "--envs-per-gpu 512 --frame 300 --spp 100" 20 > gpurun_out/run_profiles_r04_ref300.log 2>&1; echo "ref300 $(grep -c 'rc=0' gpurun_out/run_profiles_r04_ref300.log)"
timeout -k 10 300 python bench.py --no-pmc --no-cpu-baseline --envs-per-gpu 512 --frame 300 --spp 100 --steps 20 --warmup 3 > gpurun_out/r04_bench_ref300_strip.json; tail -c 300 gpurun_out/r04_bench_ref300_strip.json
