#!/bin/bash
# two pixel sets per thread at 8 waves / SIMD against three at 7 (the committed kernel)
set -u
OUT=gpurun_out/r03_s2; mkdir -p $OUT
for l in s2c1 s2c0; do
  REINFOCUS_HIP_LIB=$PWD/tools/lib_$l.so timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q > $OUT/parity_$l.log 2>&1; echo "parity $l rc=$? $(tail -1 $OUT/parity_$l.log)"
done
REPS=2 bash tools/ab.sh tools/lib_base.so tools/lib_s2c1.so tools/lib_s2c0.so 2>&1 | tee $OUT/ab.log
AB_ARGS="--envs-per-gpu 512 --frame 300 --spp 100" REPS=1 bash tools/ab.sh tools/lib_base.so tools/lib_s2c1.so tools/lib_s2c0.so 2>&1 | tee $OUT/ab300.log
