#!/bin/bash
# tail parameters again, with the workers at priority 1
set -u
OUT=gpurun_out/r03_prio; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc $(tail -1 $OUT/pytest.log)"
[ $rc -eq 0 ] || { tail -30 $OUT/pytest.log; exit $rc; }
REPS=2 bash tools/ab.sh tools/lib_base.so tools/lib_r1s3.so tools/lib_r1s1.so tools/lib_trm128.so tools/lib_trm32.so tools/lib_disc64.so tools/lib_r1d2.so 2>&1 | tee $OUT/ab3.log
