#!/bin/bash
# round 4, call 11: overlapped step with the second stream at normal priority
set -u
OUT=gpurun_out/r04_k; mkdir -p $OUT; rm -f $OUT/*
C4="--no-pmc --no-cpu-baseline --envs-per-gpu 128 --frame 512 --spp 64 --steps 20 --warmup 3"
for rep in 1 2; do
for pr in 0 1; do
  REINFOCUS_ENV_OVERLAP_PRIO=$pr REINFOCUS_ENV_OVERLAP=1 timeout -k 10 300 python bench.py $C4 > $OUT/c4_prio${pr}_$rep.json 2>> $OUT/err.log
  REINFOCUS_ENV_OVERLAP_PRIO=$pr REINFOCUS_ENV_OVERLAP=1 timeout -k 10 300 python bench.py --no-pmc --no-cpu-baseline --envs-per-gpu 1024 --frame 256 --spp 16 --steps 20 --warmup 2 > $OUT/e1024_prio${pr}_$rep.json 2>> $OUT/err.log
done; done
REINFOCUS_ENV_OVERLAP_PRIO=0 REINFOCUS_ENV_OVERLAP=1 timeout -k 10 300 python bench.py --no-pmc --no-cpu-baseline --steps 10 --warmup 2 > $OUT/head_prio0.json 2>> $OUT/err.log
for f in $OUT/*.json; do python -c "
import json,sys
ls=[l for l in open('$f') if l.startswith('{')]
if not ls: print('$f', 'no line'); sys.exit()
d=json.loads(ls[-1]); print('$f', round(d['value'],1), round(d['ms_per_step'],3), d.get('roofline',{}).get('avg_launch_ms'), d.get('roofline',{}).get('launches'))"; done
