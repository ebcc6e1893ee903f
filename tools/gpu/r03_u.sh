#!/bin/bash
# round 3, GPU call U: long randomised parity soaks on the final build
set -u
OUT=gpurun_out/r03_u; mkdir -p $OUT
timeout -k 10 420 python tests/soak_render.py 600 11 > $OUT/soak_render.txt 2>&1; echo "soak_render rc=$? $(tail -1 $OUT/soak_render.txt)"
timeout -k 10 300 python tests/soak_general.py 500 11 > $OUT/soak_general.txt 2>&1; echo "soak_general rc=$? $(tail -1 $OUT/soak_general.txt)"
timeout -k 10 300 python tools/soak_env.py 250 11 > $OUT/soak_env.txt 2>&1; echo "soak_env rc=$? $(tail -1 $OUT/soak_env.txt)"
