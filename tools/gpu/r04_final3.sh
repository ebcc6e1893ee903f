#!/bin/bash
# round 4, final check on the committed tree: what the driver runs -- the GPU suite, smoke(), the default bench line
set -u
OUT=gpurun_out/r04_final3; mkdir -p $OUT; rm -f $OUT/*
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -n 2 $OUT/pytest.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -n 1 $OUT/smoke.log
timeout -k 10 600 python bench.py > $OUT/bench_default.json 2> $OUT/err.log; echo "bench rc=$?"
python -c "
import json; d=json.loads([l for l in open('$OUT/bench_default.json') if l.startswith('{')][-1]); print(round(d['value'],1), round(d['ms_per_step'],3), d['roofline']['frac'], d['roofline']['traffic'], d['cpu_baseline']['value'], d['devices'])"
