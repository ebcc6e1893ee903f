#!/bin/bash
# round 4, call 25: the env step with ONE render launch (two-pass blocks for the re-rendered slots) and one focus launch
set -u
OUT=gpurun_out/r04_y; mkdir -p $OUT; rm -f $OUT/*
timeout -k 10 900 python -m pytest tests/test_gpu_environment.py tests/test_gpu_strategy_cases.py tests/test_gpu_full_size.py tests/test_gpu_bench.py -x -q -m gpu 2>&1 | tail -n 3 || exit 1
run() { # name, fused, args...
  local name=$1 fused=$2; shift 2
  REINFOCUS_ENV_FUSED=$fused timeout -k 10 300 python bench.py --no-pmc --no-cpu-baseline "$@" > $OUT/${name}_fused$fused.json 2>> $OUT/err.log
  python - "$OUT/${name}_fused$fused.json" <<'PY'
import json,sys
ls=[l for l in open(sys.argv[1]) if l.startswith('{')]
if not ls: print(sys.argv[1], 'no line'); sys.exit()
d=json.loads(ls[-1]); r=d.get('roofline',{})
print(sys.argv[1], round(d['value'],1), 'env-steps/s', round(d['ms_per_step'],3), 'ms', r.get('kernel'), r.get('avg_launch_ms'), r.get('launches'), flush=True)
PY
}
for rep in 1 2; do for fused in 0 1; do
  run c4_$rep $fused --envs-per-gpu 128 --frame 512 --spp 64 --steps 20 --warmup 3
  run head_$rep $fused --steps 12 --warmup 2
  run e1024_$rep $fused --envs-per-gpu 1024 --frame 256 --spp 16 --steps 20 --warmup 2
  run ref300_$rep $fused --envs-per-gpu 512 --frame 300 --spp 100 --steps 8 --warmup 2
  run c1_$rep $fused --envs-per-gpu 256 --frame 128 --spp 4 --steps 2000 --warmup 20 --no-kernel-timing
done; done
