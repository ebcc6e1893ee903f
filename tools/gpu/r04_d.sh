#!/bin/bash
# round 4, call 4: the dense pass with in-wave loops (REINFOCUS_GENERAL_DENSE=2): parity and throughput
set -u
OUT=gpurun_out/r04_d; mkdir -p $OUT; rm -f $OUT/*
REINFOCUS_GENERAL_DENSE=2 timeout -k 10 600 python -m pytest tests/test_gpu_general.py tests/test_gpu_notebook_figures.py -x -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
for rep in 1 2; do
  echo "== literal"; REINFOCUS_GENERAL_DENSE=0 timeout -k 10 300 python tools/bench_general.py 256 256 16 || exit 1
  echo "== dense in-wave"; REINFOCUS_GENERAL_DENSE=2 timeout -k 10 300 python tools/bench_general.py 256 256 16 || exit 1
done 2>&1 | tee $OUT/bench_general.txt
echo "== 300 px / 100 spp"; REINFOCUS_GENERAL_DENSE=2 timeout -k 10 300 python tools/bench_general.py 64 300 100 2>&1 | tee $OUT/bench_general_300_dense_inwave.txt
