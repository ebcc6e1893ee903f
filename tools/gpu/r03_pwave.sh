#!/bin/bash
# list slots by one LDS atomic per wave and call (ranks from v_mbcnt on the masks) against one atomic per straggler
set -u
OUT=gpurun_out/r03_pwave; mkdir -p $OUT
REINFOCUS_HIP_LIB=$PWD/tools/lib_pwave.so timeout -k 10 600 python tests/soak_render.py 40 > $OUT/soak.log 2>&1; echo "soak (pwave) rc=$? $(tail -1 $OUT/soak.log)"
REPS=3 bash tools/ab.sh tools/lib_base.so tools/lib_pwave.so 2>&1 | tee $OUT/ab.log
AB_ARGS="--envs-per-gpu 512 --frame 300 --spp 100" REPS=2 bash tools/ab.sh tools/lib_base.so tools/lib_pwave.so 2>&1 | tee $OUT/ab300.log
