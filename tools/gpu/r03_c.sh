#!/bin/bash
# round 3, GPU call C: co-issue matrix, full GPU test suite on the bitop3 build, default bench line
set -u
OUT=gpurun_out/r03_c; mkdir -p $OUT
./tools/ubench/pairbench 2.38 8 > $OUT/pairbench_8w.txt 2>&1; echo "pairbench rc=$?"
./tools/ubench/pairbench 2.38 2 > $OUT/pairbench_2w.txt 2>&1; echo "pairbench rc=$?"
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$? $(tail -1 $OUT/pytest.log)"
timeout -k 10 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"; cut -c1-600 $OUT/bench_default.json
