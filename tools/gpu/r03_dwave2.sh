#!/bin/bash
set -u
OUT=gpurun_out/r03_dwave; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/pytest2.log 2>&1; rc=$?; echo "pytest rc=$rc $(tail -1 $OUT/pytest2.log)"
[ $rc -eq 0 ] || { tail -30 $OUT/pytest2.log; exit $rc; }
timeout -k 10 900 python tests/soak_render.py 150 11 > $OUT/soak.log 2>&1; echo "soak rc=$? $(tail -1 $OUT/soak.log)"
REPS=2 bash tools/ab.sh tools/lib_base.so tools/lib_new.so 2>&1 | tee $OUT/ab2.log
AB_ARGS="--envs-per-gpu 512 --frame 300 --spp 100" REPS=2 bash tools/ab.sh tools/lib_base.so tools/lib_new.so 2>&1 | tee -a $OUT/ab2.log
AB_ARGS="--envs-per-gpu 256 --frame 128 --spp 4" REPS=2 bash tools/ab.sh tools/lib_base.so tools/lib_new.so 2>&1 | tee -a $OUT/ab2.log
