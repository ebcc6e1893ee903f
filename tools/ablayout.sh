#!/bin/bash
# Development tool: render-kernel tile layouts (REINFOCUS_TILE_LAYOUT 0..3) across frame sizes.
for cfg in "1024 128 16 20" "4096 256 16 8" "512 300 100 6" "768 384 32 6" "1024 512 64 4" "256 600 100 4"; do set -- $cfg
  for l in 0 1 2 3 4; do
    REINFOCUS_TILE_LAYOUT=$l timeout -k 10 200 python bench.py --no-cpu-baseline --envs-per-gpu $1 --frame $2 --spp $3 --steps $4 --warmup 2 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('frame $2 layout $l', round(d['value']), round(d['roofline']['samples_per_s']/1e9,1), flush=True)"
  done
done
