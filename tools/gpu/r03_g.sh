#!/bin/bash
# round 3, GPU call G: full GPU suite on the new default; retuning sweep of the cooperative parameters
set -u
OUT=gpurun_out/r03_g; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$? $(tail -1 $OUT/pytest.log)"
REPS=2 bash tools/ab.sh tools/lib_base.so tools/lib_cl1.so tools/lib_r1s3.so tools/lib_r1s1.so tools/lib_trmin32.so tools/lib_trmin128.so tools/lib_disc2.so tools/lib_trips2.so 2>&1 | tee $OUT/ab.log
