#!/bin/bash
# two-wave blocks (128 threads, tile 64 x 6) for render_kernel_coop2 against four-wave blocks
set -u
OUT=gpurun_out/r03_nw2; mkdir -p $OUT
REINFOCUS_HIP_LIB=$PWD/tools/lib_nw2.so timeout -k 10 600 python tests/soak_render.py 60 3 > $OUT/soak.log 2>&1; rc=$?; echo "soak (nw2) rc=$rc $(tail -1 $OUT/soak.log)"
[ $rc -eq 0 ] || { tail -5 $OUT/soak.log; exit $rc; }
REPS=2 bash tools/ab.sh tools/lib_base.so tools/lib_nw2.so 2>&1 | tee $OUT/ab.log
for l in 2 4; do echo "-- nw2, REINFOCUS_TILE_LAYOUT=$l"; REINFOCUS_TILE_LAYOUT=$l REPS=1 bash tools/ab.sh tools/lib_nw2.so 2>&1 | tee -a $OUT/ab.log; done
AB_ARGS="--envs-per-gpu 512 --frame 300 --spp 100" REPS=1 bash tools/ab.sh tools/lib_base.so tools/lib_nw2.so 2>&1 | tee -a $OUT/ab.log
AB_ARGS="--envs-per-gpu 128 --frame 512 --spp 64" REPS=1 bash tools/ab.sh tools/lib_base.so tools/lib_nw2.so 2>&1 | tee -a $OUT/ab.log
