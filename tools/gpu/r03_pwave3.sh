#!/bin/bash
# list slots per wave for the power-of-two instances, per straggler for the others: parity, then both against all-per-straggler
set -u
OUT=gpurun_out/r03_pwave; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc $(tail -1 $OUT/pytest.log)"
[ $rc -eq 0 ] || { tail -30 $OUT/pytest.log; exit $rc; }
timeout -k 10 600 python tests/soak_render.py 60 5 > $OUT/soak2.log 2>&1; echo "soak rc=$? $(tail -1 $OUT/soak2.log)"
REPS=2 bash tools/ab.sh tools/lib_base.so tools/lib_new.so 2>&1 | tee $OUT/ab_final.log
AB_ARGS="--envs-per-gpu 512 --frame 300 --spp 100" REPS=2 bash tools/ab.sh tools/lib_base.so tools/lib_new.so 2>&1 | tee -a $OUT/ab_final.log
