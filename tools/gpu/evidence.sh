#!/bin/bash
# The evidence the profiles/ directory holds per round, as two GPU-box jobs (each one `gpurun --timeout 1200` call):
#   bash tools/gpu/evidence.sh profiles <tag>   rocprofv3 kernel stats + PMC passes of every single-GPU configuration and of the general
#                                               renderer's kernels  (then, in the container: python profiles/summarize.py <tag>[_...])
#   bash tools/gpu/evidence.sh lines <tag>      the GPU suite + smoke, clean bench lines of every single-GPU configuration, the three
#                                               routes of the boundary, multi-rank rehearsals on device 0, randomised parity soaks
set -u
job=${1:?profiles|lines}; tag=${2:?tag, e.g. r05}
if [ "$job" = profiles ]; then
  mkdir -p gpurun_out
  run() { bash profiles/run_profiles.sh "$@" > gpurun_out/run_profiles_$1.log 2>&1; echo "$1: $(grep -c 'rc=0' gpurun_out/run_profiles_$1.log) passes ok"; }
  run $tag "" 50
  run ${tag}_ref300 "--envs-per-gpu 512 --frame 300 --spp 100" 20
  run ${tag}_c4 "--envs-per-gpu 128 --frame 512 --spp 64" 20
  run ${tag}_c1 "--envs-per-gpu 256 --frame 128 --spp 4" 200
  run ${tag}_ref8 "--envs-per-gpu 8 --frame 300 --spp 100" 100   # the reference's training shape: render_kernel_wave
  run ${tag}_env1 "--envs-per-gpu 1 --frame 300 --spp 100" 200   # the reference's default environment: render_kernel
  for scene in one_rect one_sphere two_sphere mixed; do run ${tag}_general_$scene "256 256 16 --scene $scene" 0 tools/bench_general.py; done
  exit 0
fi
OUT=gpurun_out/${tag}_lines; mkdir -p $OUT; rm -f $OUT/*
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -n 2 $OUT/pytest.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 2
B="python bench.py --no-pmc --no-cpu-baseline"
python bench.py --steps 50 --warmup 5 > $OUT/bench_default.json 2> $OUT/err.log; echo "default rc=$?"
python tools/bench_general.py 256 256 16 > $OUT/bench_general.txt 2>&1
python tools/bench_general.py 64 300 100 > $OUT/bench_general_300.txt 2>&1
$B --envs-per-gpu 256 --frame 128 --spp 4 --steps 500 --warmup 20 > $OUT/bench_c1_events.json 2>> $OUT/err.log
$B --no-kernel-timing --envs-per-gpu 256 --frame 128 --spp 4 --steps 2000 --warmup 20 > $OUT/bench_c1_graph.json 2>> $OUT/err.log
$B --envs-per-gpu 128 --frame 512 --spp 64 --steps 20 --warmup 3 > $OUT/bench_c4_share.json 2>> $OUT/err.log
$B --envs-per-gpu 1024 --frame 512 --spp 64 --steps 5 --warmup 1 > $OUT/bench_c4_whole_one_gpu.json 2>> $OUT/err.log
$B --envs-per-gpu 512 --frame 300 --spp 100 --steps 20 --warmup 3 > $OUT/bench_ref300.json 2>> $OUT/err.log
$B --envs-per-gpu 1 --frame 64 --spp 1 --steps 3000 --warmup 20 --no-kernel-timing > $OUT/bench_c0_gpu.json 2>> $OUT/err.log
$B --envs-per-gpu 1 --frame 300 --spp 100 --steps 500 --warmup 20 --no-kernel-timing > $OUT/bench_default_env.json 2>> $OUT/err.log
$B --envs-per-gpu 8 --frame 300 --spp 100 --steps 300 --warmup 20 --no-kernel-timing > $OUT/bench_ref8.json 2>> $OUT/err.log
$B --envs-per-gpu 16 --frame 256 --spp 16 --steps 500 --warmup 20 --no-kernel-timing > $OUT/bench_16x256.json 2>> $OUT/err.log
for e in device host literal; do $B --steps 20 --warmup 3 --env $e > $OUT/route_$e.json 2>> $OUT/err.log; done
export REINFOCUS_BENCH_DEVICE=0
timeout -k 10 600 $B --gpus 8 --sharded-env --steps 10 --warmup 2 > $OUT/rehearsal_sharded8_one_device.json 2>> $OUT/err.log; echo "sharded8 rc=$?"
timeout -k 10 600 $B --gpus 6 --steps 10 --warmup 2 > $OUT/rehearsal_ranks6_one_device.json 2>> $OUT/err.log; echo "ranks6 rc=$?"
timeout -k 10 600 $B --gpus 2 --steps 10 --warmup 2 > $OUT/rehearsal_ranks2_one_device.json 2>> $OUT/err.log; echo "ranks2 rc=$?"
timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 4 --steps 10 --warmup 2 --no-cpu-baseline --no-pmc > $OUT/rehearsal_torchrun4_one_device.json 2>> $OUT/err.log; echo "torchrun4 rc=$?"
unset REINFOCUS_BENCH_DEVICE
for f in $OUT/bench_*.json $OUT/route_*.json $OUT/rehearsal_*.json; do python -c "
import json; d=json.loads([l for l in open('$f') if l.startswith('{')][-1]); print('$f', d['n_gpus'], round(d['value'],1), round(d['ms_per_step'],4), d['config']['workload'][:60], len(d['devices']), d['config'].get('host_gather_bytes_per_step'))"; done
cat $OUT/bench_general.txt $OUT/bench_general_300.txt
{ echo "## tests/soak/soak_render.py 600 25"; timeout -k 10 900 python tests/soak/soak_render.py 600 25 2>&1 | tail -n 2
  for sets in w3 w2 3; do echo "## REINFOCUS_RENDER_SETS=$sets tests/soak/soak_render.py 400 26"; REINFOCUS_RENDER_SETS=$sets timeout -k 10 900 python tests/soak/soak_render.py 400 26 2>&1 | tail -n 1; done
  echo "## tests/soak/soak_focus.py 300 27"; timeout -k 10 900 python tests/soak/soak_focus.py 300 27 2>&1 | tail -n 1
  echo "## tests/soak/soak_general.py 800 25"; timeout -k 10 600 python tests/soak/soak_general.py 800 25 2>&1 | tail -n 2
  echo "## tools/soak_env.py 250 25"; timeout -k 10 600 python tools/soak_env.py 250 25 2>&1 | tail -n 2; } > $OUT/soaks.txt 2>&1
cat $OUT/soaks.txt
