#!/bin/bash
# round 3, GPU call F: parity + A/B of -fno-slp-vectorize and the magic u64 -> f64 conversion
set -u
OUT=gpurun_out/r03_f; mkdir -p $OUT
for v in noslp both; do
  REINFOCUS_HIP_LIB=$PWD/tools/lib_$v.so timeout -k 10 600 python -m pytest tests -m gpu -x -q > $OUT/pytest_$v.log 2>&1; echo "pytest $v rc=$? $(tail -1 $OUT/pytest_$v.log)"
done
REPS=3 bash tools/ab.sh tools/lib_base.so tools/lib_noslp.so tools/lib_both.so 2>&1 | tee $OUT/ab.log
AB_ARGS="--envs-per-gpu 512 --frame 300 --spp 100" REPS=2 bash tools/ab.sh tools/lib_base.so tools/lib_noslp.so tools/lib_both.so 2>&1 | tee $OUT/ab300.log
