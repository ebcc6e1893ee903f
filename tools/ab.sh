#!/bin/bash
# Development tool: A/B builds of libreinfocus_hip.so on the GPU box (headline configuration unless
# AB_ARGS says otherwise).  Each variant is benched REPS times (default 2), interleaved.
# usage: [AB_ARGS="--envs-per-gpu 512 --frame 300 --spp 100"] [REPS=2] bash tools/ab.sh tools/lib_a.so tools/lib_b.so ...
REPS=${REPS:-2}
for rep in $(seq $REPS); do
  for lib in "$@"; do
    REINFOCUS_HIP_LIB=$PWD/$lib timeout -k 10 200 python bench.py --no-cpu-baseline --no-pmc --steps 10 --warmup 2 $AB_ARGS | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$lib', round(d['value']), round(d['roofline']['samples_per_s']/1e9,2), d['roofline']['kernel'], flush=True)" || exit 1
  done
done
