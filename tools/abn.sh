#!/bin/bash
# Development tool: like ab.sh, at 512 envs x 300^2 x 100 spp (the reference's default frame).
cp reinfocus_amd/libreinfocus_hip.so /tmp/lib_orig.so
for rep in 1 2; do for lib in "$@"; do cp $lib reinfocus_amd/libreinfocus_hip.so; timeout -k 10 200 python bench.py --no-cpu-baseline --no-pmc --envs-per-gpu 512 --frame 300 --spp 100 --steps 10 --warmup 2 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib', round(d['value']), round(d['roofline']['samples_per_s']/1e9,1), flush=True)"; done; done
cp /tmp/lib_orig.so reinfocus_amd/libreinfocus_hip.so
