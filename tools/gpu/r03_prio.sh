#!/bin/bash
# worker waves at raised priority inside the cooperative tails
set -u
OUT=gpurun_out/r03_prio; mkdir -p $OUT
REPS=3 bash tools/ab.sh tools/lib_base.so tools/lib_prio1.so tools/lib_prio2.so tools/lib_prio1r2.so 2>&1 | tee $OUT/ab2.log
AB_ARGS="--envs-per-gpu 512 --frame 300 --spp 100" REPS=2 bash tools/ab.sh tools/lib_base.so tools/lib_prio1.so tools/lib_prio1r2.so 2>&1 | tee $OUT/ab300.log
