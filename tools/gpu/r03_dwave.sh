#!/bin/bash
# disc tails inside the wave (no block barrier) against the block-cooperative form
set -u
OUT=gpurun_out/r03_dwave; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc $(tail -1 $OUT/pytest.log)"
[ $rc -eq 0 ] || { tail -30 $OUT/pytest.log; exit $rc; }
REPS=3 bash tools/ab.sh tools/lib_base.so tools/lib_new.so 2>&1 | tee $OUT/ab.log
AB_ARGS="--envs-per-gpu 512 --frame 300 --spp 100" REPS=2 bash tools/ab.sh tools/lib_base.so tools/lib_new.so 2>&1 | tee -a $OUT/ab.log
AB_ARGS="--envs-per-gpu 128 --frame 512 --spp 64" REPS=2 bash tools/ab.sh tools/lib_base.so tools/lib_new.so 2>&1 | tee -a $OUT/ab.log
