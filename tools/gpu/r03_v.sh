#!/bin/bash
set -u
OUT=gpurun_out/r03_v; mkdir -p $OUT
REINFOCUS_HIP_LIB=$PWD/tools/lib_patom.so timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_patom.log 2>&1; echo "pytest patom rc=$? $(tail -1 $OUT/pytest_patom.log)"
REPS=3 bash tools/ab.sh tools/lib_base.so tools/lib_patom.so 2>&1 | tee $OUT/ab.log
AB_ARGS="--envs-per-gpu 512 --frame 300 --spp 100" REPS=2 bash tools/ab.sh tools/lib_base.so tools/lib_patom.so 2>&1 | tee $OUT/ab300.log
