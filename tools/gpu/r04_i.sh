#!/bin/bash
# round 4, call 9: the literal drop-in route at the headline configuration
set -u
OUT=gpurun_out/r04_h; mkdir -p $OUT
timeout -k 10 600 python bench.py --no-pmc --no-cpu-baseline --steps 10 --warmup 2 --env literal > $OUT/bench_literal.json 2> $OUT/err_literal.log; echo "literal rc=$?"
grep -v "amdgpu.ids" $OUT/err_literal.log | tail -n 5
python -c "
import json
d=json.loads([l for l in open('$OUT/bench_literal.json') if l.startswith('{')][-1]); print(round(d['value'],1), round(d['ms_per_step'],3), d['config']['env_glue'], d['pcie_bytes_per_step_per_gpu'], d['kernel_time_frac_of_wall'])"
