#!/bin/bash
# round 4, call 8: full GPU suite on the current build; the literal drop-in route measured next to the resident ones;
# the device table of a rank and of the sharded object
set -u
OUT=gpurun_out/r04_h; mkdir -p $OUT; rm -f $OUT/*
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -n 3 $OUT/pytest.log
for e in device host literal; do
  timeout -k 10 600 python bench.py --no-pmc --no-cpu-baseline --steps 10 --warmup 2 --env $e > $OUT/bench_$e.json 2>> $OUT/err.log; echo "$e rc=$?"
done
REINFOCUS_BENCH_DEVICE=0 timeout -k 10 600 python bench.py --gpus 2 --sharded-env --envs-per-gpu 1024 --steps 5 --warmup 1 --no-cpu-baseline --no-pmc > $OUT/bench_sharded2.json 2>> $OUT/err.log; echo "sharded rc=$?"
REINFOCUS_BENCH_DEVICE=0 timeout -k 10 600 python bench.py --gpus 2 --envs-per-gpu 1024 --steps 5 --warmup 1 --no-cpu-baseline --no-pmc > $OUT/bench_ranks2.json 2>> $OUT/err.log; echo "ranks2 rc=$?"
timeout -k 10 600 python bench.py --gpus 2 --envs-per-gpu 64 --steps 2 --warmup 1 --no-cpu-baseline --no-pmc > $OUT/bench_ranks2_same_gpu.json 2>> $OUT/err_same.log; echo "ranks2 on one GPU without the rehearsal switch rc=$? (must fail)"
tail -n 3 $OUT/err_same.log
for f in $OUT/bench_*.json; do python -c "
import json,sys
ls=[l for l in open('$f') if l.startswith('{')]
if not ls: print('$f', 'no line'); sys.exit()
d=json.loads(ls[-1]); print('$f', d['n_gpus'], round(d['value'],1), round(d['ms_per_step'],3), d['config']['env_glue'][:40], d['devices'])"; done
nproc; cat /sys/devices/system/node/node*/cpulist | head; tail -n 5 $OUT/err.log
