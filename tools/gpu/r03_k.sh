#!/bin/bash
set -u
OUT=gpurun_out/r03_k; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$? $(tail -1 $OUT/pytest.log)"
REPS=3 bash tools/ab.sh tools/lib_base.so tools/lib_chk.so 2>&1 | tee $OUT/ab.log
