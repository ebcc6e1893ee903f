#!/bin/bash
# round 4, call 37: the rare exact block of the sphere attempts one component at a time (no scratch on 46 of 48 instances): A/B
set -u
{ REPS=3 bash tools/ab.sh tools/lib_old.so tools/lib_seq.so
AB_ARGS="--envs-per-gpu 512 --frame 300 --spp 100" REPS=3 bash tools/ab.sh tools/lib_old.so tools/lib_seq.so
AB_ARGS="--envs-per-gpu 128 --frame 512 --spp 64" REPS=2 bash tools/ab.sh tools/lib_old.so tools/lib_seq.so
AB_ARGS="--envs-per-gpu 1024 --frame 128 --spp 16" REPS=2 bash tools/ab.sh tools/lib_old.so tools/lib_seq.so
AB_ARGS="--envs-per-gpu 256 --frame 600 --spp 16" REPS=2 bash tools/ab.sh tools/lib_old.so tools/lib_seq.so; } 2>&1 | tee gpurun_out/r04_ae.txt
