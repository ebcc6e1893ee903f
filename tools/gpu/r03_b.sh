#!/bin/bash
# round 3, GPU call B: instruction-order microbenchmark, new environment parity tests, A/B of RF_BITOP3
set -u
OUT=gpurun_out/r03_b; mkdir -p $OUT
./tools/ubench/seqbench 2.38 > $OUT/seqbench.txt 2>&1; echo "seqbench rc=$?"; cat $OUT/seqbench.txt
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $OUT/pytest_base.log 2>&1; echo "pytest base rc=$? $(tail -1 $OUT/pytest_base.log)"
REINFOCUS_HIP_LIB=$PWD/tools/lib_b3.so timeout -k 10 600 python -m pytest tests -m gpu -x -q > $OUT/pytest_b3.log 2>&1; echo "pytest b3 rc=$? $(tail -1 $OUT/pytest_b3.log)"
REPS=3 bash tools/ab.sh tools/lib_base.so tools/lib_b3.so 2>&1 | tee $OUT/ab.log
