#!/bin/bash
# general renderer: power-of-two instance (float32 coordinates) vs the previous build, parity first
set -u
OUT=gpurun_out/r03_gp; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc $(tail -1 $OUT/pytest.log)"
[ $rc -eq 0 ] || exit $rc
for i in 1 2; do bash tools/ab_general.sh tools/lib_base.so tools/lib_gpow2.so tools/lib_gpow2o6.so 2>&1 | tee -a $OUT/ab_general.log || exit 1; done
timeout -k 10 600 python tests/soak_general.py 150 3 > $OUT/soak_general.log 2>&1; echo "soak rc=$? $(tail -1 $OUT/soak_general.log)"
