#!/bin/bash
# round 3, GPU call L: randomised parity soaks on the final build; general-kernel occupancy variants
set -u
OUT=gpurun_out/r03_l; mkdir -p $OUT
timeout -k 10 500 python tests/soak_render.py 150 3 > $OUT/soak_render.txt 2>&1; echo "soak_render rc=$? $(tail -1 $OUT/soak_render.txt)"
timeout -k 10 300 python tests/soak_general.py 150 3 > $OUT/soak_general.txt 2>&1; echo "soak_general rc=$? $(tail -1 $OUT/soak_general.txt)"
timeout -k 10 300 python tools/soak_env.py 60 3 > $OUT/soak_env.txt 2>&1; echo "soak_env rc=$? $(tail -1 $OUT/soak_env.txt)"
bash tools/ab_general.sh tools/lib_gocc4.so tools/lib_gocc5.so tools/lib_gocc6.so > $OUT/ab_general_occ.txt 2>&1; cat $OUT/ab_general_occ.txt
