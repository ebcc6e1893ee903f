#!/bin/bash
# after the mask form: atomics of all sets first, scalar straggler counts, cheaper geometry; re-tuning of the tail parameters
set -u
OUT=gpurun_out/r03_masks2; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc $(tail -1 $OUT/pytest.log)"
[ $rc -eq 0 ] || { tail -30 $OUT/pytest.log; exit $rc; }
REPS=2 bash tools/ab.sh tools/lib_base.so tools/lib_new2.so tools/lib_r1s3.so tools/lib_r1s1.so tools/lib_trm128.so tools/lib_trm32.so tools/lib_disc64.so 2>&1 | tee $OUT/ab.log
AB_ARGS="--envs-per-gpu 512 --frame 300 --spp 100" REPS=1 bash tools/ab.sh tools/lib_base.so tools/lib_new2.so tools/lib_r1s3.so tools/lib_trm128.so 2>&1 | tee $OUT/ab300.log
