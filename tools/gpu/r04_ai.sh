#!/bin/bash
# round 4, call 42: strip kernel as shipped: full GPU suite, soaks, bench lines
set -u -o pipefail
timeout -k 10 1000 python -m pytest tests -x -q -m gpu 2>&1 | tail -n 3 || exit 1
timeout -k 10 900 python tests/soak_render.py 400 16 2>&1 | tail -n 1 || exit 1
timeout -k 10 600 python tools/soak_env.py 150 43 2>&1 | tail -n 1 || exit 1
for rep in 1 2; do
timeout -k 10 300 python bench.py --no-pmc --no-cpu-baseline --envs-per-gpu 512 --frame 300 --spp 100 --steps 20 --warmup 3 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('ref300', round(d['value'],1), round(d['roofline']['samples_per_s']/1e9,2), d['roofline']['kernel'], flush=True)"
done
