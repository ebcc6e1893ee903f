#!/bin/bash
set -u
OUT=gpurun_out/r03_pwave; mkdir -p $OUT
AB_ARGS="--envs-per-gpu 128 --frame 512 --spp 64" REPS=2 bash tools/ab.sh tools/lib_base.so tools/lib_pwave.so 2>&1 | tee $OUT/ab512.log
AB_ARGS="--envs-per-gpu 256 --frame 128 --spp 4" REPS=2 bash tools/ab.sh tools/lib_base.so tools/lib_pwave.so 2>&1 | tee $OUT/ab128.log
AB_ARGS="--envs-per-gpu 512 --frame 384 --spp 32" REPS=2 bash tools/ab.sh tools/lib_base.so tools/lib_pwave.so 2>&1 | tee $OUT/ab384.log
AB_ARGS="--envs-per-gpu 256 --frame 600 --spp 32" REPS=2 bash tools/ab.sh tools/lib_base.so tools/lib_pwave.so 2>&1 | tee $OUT/ab600.log
