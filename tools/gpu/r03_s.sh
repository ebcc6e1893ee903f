#!/bin/bash
set -u
OUT=gpurun_out/r03_s; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$? $(tail -1 $OUT/pytest.log)"
python tools/bench_general.py 256 256 16 > $OUT/bench_general.txt 2>&1; cat $OUT/bench_general.txt
timeout -k 10 300 python tests/soak_general.py 80 5 > $OUT/soak_general.txt 2>&1; echo "soak_general rc=$? $(tail -1 $OUT/soak_general.txt)"
