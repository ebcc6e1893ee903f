#!/bin/bash
# round 4, final evidence after the fused env step, call 2: full GPU suite, smoke, clean bench lines of every single-GPU configuration,
# the three routes of the boundary, multi-shard rehearsals on device 0, randomised parity soaks
set -u
OUT=gpurun_out/r04_final6; mkdir -p $OUT; rm -f $OUT/*
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -n 2 $OUT/pytest.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 1
python bench.py --steps 50 --warmup 5 > $OUT/bench_default.json 2> $OUT/err.log; echo "default rc=$?"
python tools/bench_general.py 256 256 16 > $OUT/bench_general.txt 2>&1
python bench.py --no-pmc --no-cpu-baseline --envs-per-gpu 256 --frame 128 --spp 4 --steps 500 --warmup 20 > $OUT/bench_c1_events.json 2>> $OUT/err.log
python bench.py --no-pmc --no-cpu-baseline --no-kernel-timing --envs-per-gpu 256 --frame 128 --spp 4 --steps 2000 --warmup 20 > $OUT/bench_c1_graph.json 2>> $OUT/err.log
python bench.py --no-pmc --no-cpu-baseline --envs-per-gpu 128 --frame 512 --spp 64 --steps 20 --warmup 3 > $OUT/bench_c4_share.json 2>> $OUT/err.log
python bench.py --no-pmc --no-cpu-baseline --envs-per-gpu 1024 --frame 512 --spp 64 --steps 5 --warmup 1 > $OUT/bench_c4_whole_one_gpu.json 2>> $OUT/err.log
python bench.py --no-pmc --no-cpu-baseline --envs-per-gpu 512 --frame 300 --spp 100 --steps 20 --warmup 3 > $OUT/bench_ref300.json 2>> $OUT/err.log
python bench.py --no-pmc --no-cpu-baseline --envs-per-gpu 1 --frame 64 --spp 1 --steps 3000 --warmup 20 --no-kernel-timing > $OUT/bench_c0_gpu.json 2>> $OUT/err.log
for e in device host literal; do python bench.py --no-pmc --no-cpu-baseline --steps 20 --warmup 3 --env $e > $OUT/route_$e.json 2>> $OUT/err.log; done
REINFOCUS_BENCH_DEVICE=0 timeout -k 10 600 python bench.py --gpus 8 --sharded-env --steps 10 --warmup 2 --no-cpu-baseline --no-pmc > $OUT/rehearsal_sharded8_one_device.json 2>> $OUT/err.log; echo "sharded8 rc=$?"
REINFOCUS_BENCH_DEVICE=0 timeout -k 10 600 python bench.py --gpus 6 --steps 10 --warmup 2 --no-cpu-baseline --no-pmc > $OUT/rehearsal_ranks6_one_device.json 2>> $OUT/err.log; echo "ranks6 rc=$?"
REINFOCUS_BENCH_DEVICE=0 timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 4 --steps 10 --warmup 2 --no-cpu-baseline --no-pmc > $OUT/rehearsal_torchrun4_one_device.json 2>> $OUT/err.log; echo "torchrun4 rc=$?"
for f in $OUT/bench_*.json $OUT/route_*.json $OUT/rehearsal_*.json; do python -c "
import json; d=json.loads([l for l in open('$f') if l.startswith('{')][-1]); print('$f', d['n_gpus'], round(d['value'],1), round(d['ms_per_step'],4), d['config']['workload'][:60], len(d['devices']))"; done
cat $OUT/bench_general.txt
{ echo "## tests/soak_render.py 600 15"; timeout -k 10 900 python tests/soak_render.py 600 15 2>&1 | tail -n 3; echo "## tests/soak_general.py 500 15"; timeout -k 10 600 python tests/soak_general.py 500 15 2>&1 | tail -n 2; echo "## tools/soak_env.py 250 15"; timeout -k 10 600 python tools/soak_env.py 250 15 2>&1 | tail -n 2; } > $OUT/soaks.txt 2>&1
cat $OUT/soaks.txt
