#!/bin/bash
# round 3, GPU call P: new >65535-environment tests; timing bound on the sparse tails (capped tail trips: wrong frames, timing only)
set -u
OUT=gpurun_out/r03_p; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$? $(tail -1 $OUT/pytest.log)"
REPS=2 bash tools/ab.sh tools/lib_base.so tools/lib_s0.so tools/lib_s1.so tools/lib_s2.so tools/lib_d1.so tools/lib_d2.so tools/lib_s0d1.so 2>&1 | tee $OUT/ab_tailcap.log
AB_ARGS="--envs-per-gpu 512 --frame 300 --spp 100" REPS=1 bash tools/ab.sh tools/lib_base.so tools/lib_s0.so tools/lib_s0d1.so 2>&1 | tee $OUT/ab_tailcap300.log
