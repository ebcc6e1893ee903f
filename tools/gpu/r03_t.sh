#!/bin/bash
# round 3, GPU call T: multi-shard rehearsals on device 0 with the final build
set -u
OUT=gpurun_out/r03_t; mkdir -p $OUT
python bench.py --no-pmc --no-cpu-baseline --steps 20 --warmup 3 > $OUT/bench_one.json 2> $OUT/err.log; echo "one rc=$?"
REINFOCUS_BENCH_DEVICE=0 timeout -k 10 600 python bench.py --gpus 8 --sharded-env --steps 10 --warmup 2 --no-cpu-baseline --no-pmc > $OUT/bench_sharded8_one_device.json 2>> $OUT/err.log; echo "sharded8 rc=$?"
REINFOCUS_BENCH_DEVICE=0 timeout -k 10 600 python bench.py --gpus 6 --steps 10 --warmup 2 --no-cpu-baseline --no-pmc > $OUT/bench_ranks6_one_device.json 2>> $OUT/err.log; echo "ranks6 rc=$?"
REINFOCUS_BENCH_DEVICE=0 timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 4 --steps 10 --warmup 2 --no-cpu-baseline --no-pmc > $OUT/bench_torchrun4_one_device.json 2>> $OUT/err.log; echo "torchrun4 rc=$?"
for f in $OUT/bench_*.json; do python -c "
import json; d=json.loads([l for l in open('$f') if l.startswith('{')][-1]); print('$f', d['n_gpus'], round(d['value']), round(d['ms_per_step'],2), d['config']['workload'][-60:])"; done
