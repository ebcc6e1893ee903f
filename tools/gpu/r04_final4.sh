#!/bin/bash
# round 4, final evidence, call 4: after the node reduction of the env step -- full GPU suite, smoke, the clean lines of every configuration again
set -u
OUT=gpurun_out/r04_final4; mkdir -p $OUT; rm -f $OUT/*
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -n 2 $OUT/pytest.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 1
python bench.py --steps 50 --warmup 5 > $OUT/bench_default.json 2> $OUT/err.log; echo "default rc=$?"
python bench.py --no-pmc --no-cpu-baseline --envs-per-gpu 256 --frame 128 --spp 4 --steps 500 --warmup 20 > $OUT/bench_c1_events.json 2>> $OUT/err.log
python bench.py --no-pmc --no-cpu-baseline --no-kernel-timing --envs-per-gpu 256 --frame 128 --spp 4 --steps 2000 --warmup 20 > $OUT/bench_c1_graph.json 2>> $OUT/err.log
python bench.py --no-pmc --no-cpu-baseline --envs-per-gpu 128 --frame 512 --spp 64 --steps 20 --warmup 3 > $OUT/bench_c4_share.json 2>> $OUT/err.log
python bench.py --no-pmc --no-cpu-baseline --envs-per-gpu 512 --frame 300 --spp 100 --steps 20 --warmup 3 > $OUT/bench_ref300.json 2>> $OUT/err.log
python bench.py --no-pmc --no-cpu-baseline --envs-per-gpu 1 --frame 64 --spp 1 --steps 3000 --warmup 20 --no-kernel-timing > $OUT/bench_c0_gpu.json 2>> $OUT/err.log
for f in $OUT/bench_*.json; do python -c "
import json; d=json.loads([l for l in open('$f') if l.startswith('{')][-1]); print('$f', d['n_gpus'], round(d['value'],1), round(d['ms_per_step'],4), d['config']['workload'][:60])"; done
{ echo "## tools/soak_env.py 250 41"; timeout -k 10 600 python tools/soak_env.py 250 41 2>&1 | tail -n 1; } | tee $OUT/soak_env.txt
