#!/bin/bash
# round 3, GPU call I: general renderer profiles; clean (untraced) bench lines of every single-GPU configuration
set -u
OUT=gpurun_out/r03_i; mkdir -p $OUT
for scene in one_rect one_sphere mixed; do
  bash profiles/run_profiles.sh r03_general_$scene "256 256 16 --scene $scene" 0 tools/bench_general.py > gpurun_out/run_profiles_r03_general_$scene.log 2>&1; echo "general $scene done"
done
python tools/bench_general.py 256 256 16 > $OUT/bench_general.txt 2>&1; cat $OUT/bench_general.txt
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "default rc=$?"
python bench.py --no-pmc --no-cpu-baseline --envs-per-gpu 256 --frame 128 --spp 4 --steps 500 --warmup 20 > $OUT/bench_c1_events.json 2>> $OUT/err.log; echo "c1 events rc=$?"
python bench.py --no-pmc --no-cpu-baseline --no-kernel-timing --envs-per-gpu 256 --frame 128 --spp 4 --steps 500 --warmup 20 > $OUT/bench_c1_graph.json 2>> $OUT/err.log; echo "c1 graph rc=$?"
python bench.py --no-pmc --no-cpu-baseline --envs-per-gpu 128 --frame 512 --spp 64 --steps 20 --warmup 3 > $OUT/bench_c4_share.json 2>> $OUT/err.log; echo "c4 rc=$?"
python bench.py --no-pmc --no-cpu-baseline --envs-per-gpu 1024 --frame 512 --spp 64 --steps 5 --warmup 1 > $OUT/bench_c4_whole_one_gpu.json 2>> $OUT/err.log; echo "c4 whole rc=$?"
python bench.py --no-pmc --no-cpu-baseline --envs-per-gpu 512 --frame 300 --spp 100 --steps 20 --warmup 3 > $OUT/bench_ref300.json 2>> $OUT/err.log; echo "ref300 rc=$?"
python bench.py --no-pmc --no-cpu-baseline --envs-per-gpu 1 --frame 64 --spp 1 --steps 500 --warmup 20 --no-kernel-timing > $OUT/bench_c0_gpu.json 2>> $OUT/err.log; echo "c0 rc=$?"
for f in $OUT/bench_*.json; do python -c "
import json,sys; d=json.load(open('$f')); print('$f', round(d['value'],1), d['config']['workload'][:90])"; done
