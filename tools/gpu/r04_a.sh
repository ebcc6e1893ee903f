#!/bin/bash
# round 4, call 1: parity of the pruned + fenced kernel, cost of the fence (A/B: 0 = round 3's unordered form,
# 1 = alternating counter + B4, 2 = done-flag in LDS)
set -u
OUT=gpurun_out/r04_a; mkdir -p $OUT; rm -f $OUT/*
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
REPS=3 bash tools/ab.sh tools/lib_sf0.so tools/lib_sf1.so tools/lib_sf2.so 2>&1 | tee $OUT/ab_headline.txt
AB_ARGS="--envs-per-gpu 128 --frame 512 --spp 64" REPS=2 bash tools/ab.sh tools/lib_sf0.so tools/lib_sf1.so 2>&1 | tee $OUT/ab_c4.txt
AB_ARGS="--envs-per-gpu 256 --frame 128 --spp 4 --steps 200" REPS=2 bash tools/ab.sh tools/lib_sf0.so tools/lib_sf1.so 2>&1 | tee $OUT/ab_c1.txt
