#!/bin/bash
# round 3, GPU call E: full GPU suite (new sharded / full-size tests), default bench, multi-shard rehearsals on one device
set -u
OUT=gpurun_out/r03_e; mkdir -p $OUT
timeout -k 10 1100 python -m pytest tests -m gpu -x -q --durations=8 > $OUT/pytest.log 2>&1; echo "pytest rc=$? $(tail -1 $OUT/pytest.log)"
timeout -k 10 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"; cut -c1-300 $OUT/bench_default.json
REINFOCUS_BENCH_DEVICE=0 timeout -k 10 600 python bench.py --gpus 8 --sharded-env --steps 10 --warmup 2 --no-cpu-baseline --no-pmc > $OUT/bench_sharded8_one_device.json 2> $OUT/bench_sharded8.err; echo "sharded8 rc=$?"; cut -c1-300 $OUT/bench_sharded8_one_device.json
REINFOCUS_BENCH_DEVICE=0 timeout -k 10 600 python bench.py --gpus 6 --steps 10 --warmup 2 --no-cpu-baseline --no-pmc > $OUT/bench_ranks6_one_device.json 2> $OUT/bench_ranks6.err; echo "ranks6 rc=$?"; cut -c1-300 $OUT/bench_ranks6_one_device.json
