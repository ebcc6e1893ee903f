#!/bin/bash
# round 3, GPU call D: full GPU test suite on the current build; parity + A/B of RF_APPROX_BITS; default bench line
set -u
OUT=gpurun_out/r03_d; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$? $(tail -1 $OUT/pytest.log)"
REINFOCUS_HIP_LIB=$PWD/tools/lib_ab.so timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_ab.log 2>&1; echo "pytest ab rc=$? $(tail -1 $OUT/pytest_ab.log)"
REPS=3 bash tools/ab.sh tools/lib_base.so tools/lib_ab.so 2>&1 | tee $OUT/ab.log
AB_ARGS="--envs-per-gpu 512 --frame 300 --spp 100" REPS=2 bash tools/ab.sh tools/lib_base.so tools/lib_ab.so 2>&1 | tee $OUT/ab300.log
timeout -k 10 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"; cut -c1-400 $OUT/bench_default.json
