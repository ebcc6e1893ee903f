#!/bin/bash
# round 3, final GPU call: evidence of every single-GPU configuration on the final build
set -u
bash profiles/run_profiles.sh r03_ref300 "--envs-per-gpu 512 --frame 300 --spp 100" 20 > gpurun_out/run_profiles_r03_ref300.log 2>&1; echo "ref300 $(grep -c 'rc=0' gpurun_out/run_profiles_r03_ref300.log)"
bash profiles/run_profiles.sh r03_c4 "--envs-per-gpu 128 --frame 512 --spp 64" 20 > gpurun_out/run_profiles_r03_c4.log 2>&1; echo "c4 $(grep -c 'rc=0' gpurun_out/run_profiles_r03_c4.log)"
bash profiles/run_profiles.sh r03_c1 "--envs-per-gpu 256 --frame 128 --spp 4" 200 > gpurun_out/run_profiles_r03_c1.log 2>&1; echo "c1 $(grep -c 'rc=0' gpurun_out/run_profiles_r03_c1.log)"
for scene in one_rect one_sphere mixed; do
  bash profiles/run_profiles.sh r03_general_$scene "256 256 16 --scene $scene" 0 tools/bench_general.py > gpurun_out/run_profiles_r03_general_$scene.log 2>&1; echo "general $scene $(grep -c 'rc=0' gpurun_out/run_profiles_r03_general_$scene.log)"
done
OUT=gpurun_out/r03_final; mkdir -p $OUT
python tools/bench_general.py 256 256 16 > $OUT/bench_general.txt 2>&1
python bench.py --no-pmc --no-cpu-baseline --envs-per-gpu 256 --frame 128 --spp 4 --steps 500 --warmup 20 > $OUT/bench_c1_events.json 2>> $OUT/err.log
python bench.py --no-pmc --no-cpu-baseline --no-kernel-timing --envs-per-gpu 256 --frame 128 --spp 4 --steps 500 --warmup 20 > $OUT/bench_c1_graph.json 2>> $OUT/err.log
python bench.py --no-pmc --no-cpu-baseline --envs-per-gpu 128 --frame 512 --spp 64 --steps 20 --warmup 3 > $OUT/bench_c4_share.json 2>> $OUT/err.log
python bench.py --no-pmc --no-cpu-baseline --envs-per-gpu 1024 --frame 512 --spp 64 --steps 5 --warmup 1 > $OUT/bench_c4_whole_one_gpu.json 2>> $OUT/err.log
python bench.py --no-pmc --no-cpu-baseline --envs-per-gpu 512 --frame 300 --spp 100 --steps 20 --warmup 3 > $OUT/bench_ref300.json 2>> $OUT/err.log
python bench.py --no-pmc --no-cpu-baseline --envs-per-gpu 1 --frame 64 --spp 1 --steps 500 --warmup 20 --no-kernel-timing > $OUT/bench_c0_gpu.json 2>> $OUT/err.log
for f in $OUT/bench_*.json; do python -c "
import json,sys; d=json.load(open('$f')); print('$f', round(d['value'],1), d['config']['workload'][:80])"; done
cat $OUT/bench_general.txt
