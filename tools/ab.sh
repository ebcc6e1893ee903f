#!/bin/bash
# Development tool: A/B builds of libreinfocus_hip.so on the GPU box.
# usage: bash tools/ab.sh tools/lib_a.so tools/lib_b.so ...   (each variant is benched twice, interleaved)
set -e
cp reinfocus_amd/libreinfocus_hip.so /tmp/lib_orig.so
for rep in 1 2; do
  for lib in "$@"; do
    cp "$lib" reinfocus_amd/libreinfocus_hip.so
    timeout -k 10 200 python bench.py --no-cpu-baseline --no-pmc --steps 10 --warmup 2 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$lib', round(d['value']), round(d['roofline']['samples_per_s']/1e9,2), flush=True)"
  done
done
cp /tmp/lib_orig.so reinfocus_amd/libreinfocus_hip.so
