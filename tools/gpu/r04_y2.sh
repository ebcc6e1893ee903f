#!/bin/bash
# round 4, call 27: fused env step with pre + plan and post + reset-post merged (6 graph nodes per step)
set -u -o pipefail
OUT=gpurun_out/r04_y2; mkdir -p $OUT; rm -f $OUT/*
timeout -k 10 900 python -m pytest tests/test_gpu_environment.py tests/test_gpu_strategy_cases.py tests/test_gpu_full_size.py tests/test_gpu_bench.py -x -q -m gpu 2>&1 | tail -n 3 || exit 1
timeout -k 10 600 python tools/soak_env.py 100 37 2>&1 | tail -n 1 || exit 1
run() { # name, args...
  local name=$1; shift
  timeout -k 10 300 python bench.py --no-pmc --no-cpu-baseline "$@" > $OUT/$name.json 2>> $OUT/err.log
  python - "$OUT/$name.json" <<'PY'
import json,sys
ls=[l for l in open(sys.argv[1]) if l.startswith('{')]
if not ls: print(sys.argv[1], 'no line'); sys.exit()
d=json.loads(ls[-1]); r=d.get('roofline',{})
print(sys.argv[1], round(d['value'],1), 'env-steps/s', round(d['ms_per_step'],4), 'ms', r.get('kernel'), r.get('avg_launch_ms'), r.get('launches'), flush=True)
PY
}
for rep in 1 2 3; do
  run c1_$rep --envs-per-gpu 256 --frame 128 --spp 4 --steps 2000 --warmup 20 --no-kernel-timing
  run one_$rep --envs-per-gpu 1 --frame 64 --spp 1 --steps 3000 --warmup 20 --no-kernel-timing
done
run head --steps 12 --warmup 2
run c4 --envs-per-gpu 128 --frame 512 --spp 64 --steps 20 --warmup 3
