#!/bin/bash
set -u
OUT=gpurun_out/r03_m; mkdir -p $OUT
REPS=2 bash tools/ab.sh tools/lib_base.so tools/lib_o2.so tools/lib_nopost.so tools/lib_nomisched.so tools/lib_nounroll.so tools/lib_relaxed.so 2>&1 | tee $OUT/ab.log
