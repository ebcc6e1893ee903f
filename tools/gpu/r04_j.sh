#!/bin/bash
# round 4, call 10: the overlapped env step: parity (all four schedules, full size), then the per-GPU share of configs[4]
# with it forced on / off, with kernel events and without (graph replay when off)
set -u
OUT=gpurun_out/r04_j; mkdir -p $OUT; rm -f $OUT/*
timeout -k 10 900 python -m pytest tests/test_gpu_environment.py tests/test_gpu_full_size.py tests/test_gpu_strategy_cases.py -x -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -n 3 $OUT/pytest.log
C4="--no-pmc --no-cpu-baseline --envs-per-gpu 128 --frame 512 --spp 64 --steps 20 --warmup 3"
for rep in 1 2; do
for ov in 0 1; do
  REINFOCUS_ENV_OVERLAP=$ov timeout -k 10 300 python bench.py $C4 > $OUT/c4_ov${ov}_events_$rep.json 2>> $OUT/err.log
  REINFOCUS_ENV_OVERLAP=$ov timeout -k 10 300 python bench.py $C4 --no-kernel-timing > $OUT/c4_ov${ov}_noevents_$rep.json 2>> $OUT/err.log
done; done
for ov in 0 1; do REINFOCUS_ENV_OVERLAP=$ov timeout -k 10 300 python bench.py --no-pmc --no-cpu-baseline --steps 10 --warmup 2 > $OUT/head_ov$ov.json 2>> $OUT/err.log; done
for ov in 0 1; do REINFOCUS_ENV_OVERLAP=$ov timeout -k 10 300 python bench.py --no-pmc --no-cpu-baseline --envs-per-gpu 512 --frame 300 --spp 100 --steps 10 --warmup 2 > $OUT/ref300_ov$ov.json 2>> $OUT/err.log; done
for ov in 0 1; do REINFOCUS_ENV_OVERLAP=$ov timeout -k 10 300 python bench.py --no-pmc --no-cpu-baseline --envs-per-gpu 1024 --frame 256 --spp 16 --steps 20 --warmup 2 > $OUT/e1024_ov$ov.json 2>> $OUT/err.log; done
for f in $OUT/*.json; do python -c "
import json,sys
ls=[l for l in open('$f') if l.startswith('{')]
if not ls: print('$f', 'no line'); sys.exit()
d=json.loads(ls[-1]); print('$f', round(d['value'],1), round(d['ms_per_step'],3), d.get('roofline',{}).get('avg_launch_ms'), d.get('roofline',{}).get('launches'))"; done
grep -v "amdgpu.ids" $OUT/err.log | tail -n 5
