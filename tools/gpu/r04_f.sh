#!/bin/bash
# round 4, call 6: literal / dense cooperative / dense in-wave general renderer after the hoisting fix
set -u
OUT=gpurun_out/r04_f; mkdir -p $OUT; rm -f $OUT/*
for m in 1 2; do REINFOCUS_GENERAL_DENSE=$m timeout -k 10 600 python -m pytest tests/test_gpu_general.py -x -q -m gpu > $OUT/pytest_$m.log 2>&1; echo "pytest mode $m rc=$?"; tail -1 $OUT/pytest_$m.log; done
for rep in 1 2; do
  for m in 0 1 2; do echo "== mode $m"; REINFOCUS_GENERAL_DENSE=$m timeout -k 10 300 python tools/bench_general.py 256 256 16 || exit 1; done
done 2>&1 | tee $OUT/bench_general.txt
for m in 0 1 2; do echo "== 300 px / 100 spp mode $m"; REINFOCUS_GENERAL_DENSE=$m timeout -k 10 300 python tools/bench_general.py 64 300 100; done 2>&1 | tee $OUT/bench_general_300.txt
