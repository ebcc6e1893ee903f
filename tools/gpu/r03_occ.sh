#!/bin/bash
# does the three-pixel kernel still gain from every resident wave?  (LDS padding: 6 / 5 / 4 blocks per CU; and a
# 6-wave register budget with and without the things the 7-wave budget forced out of registers)
set -u
OUT=gpurun_out/r03_occ; mkdir -p $OUT
REPS=2 bash tools/ab.sh tools/lib_base.so tools/lib_pad6.so tools/lib_pad5.so tools/lib_pad4.so tools/lib_occ6.so tools/lib_occ6g.so tools/lib_occ6c0.so 2>&1 | tee $OUT/ab.log
