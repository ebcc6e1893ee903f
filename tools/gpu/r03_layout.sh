#!/bin/bash
# tile layouts again on the final kernel + the overflow paths of the in-wave disc tails (8-slot test build)
set -u
OUT=gpurun_out/r03_layout; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc $(tail -1 $OUT/pytest.log)"
[ $rc -eq 0 ] || { tail -30 $OUT/pytest.log; exit $rc; }
bash tools/ablayout.sh 2>&1 | tee $OUT/layout.log
