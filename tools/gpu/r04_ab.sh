#!/bin/bash
# round 4, call 31: occupancy of the single-sphere kernel
set -u
for rep in 1 2; do for lib in reinfocus_amd/libreinfocus_hip.so tools/lib_s4.so tools/lib_s5.so tools/lib_s7.so; do
  echo "== $lib"; REINFOCUS_HIP_LIB=$PWD/$lib timeout -k 10 300 python tools/bench_general.py 256 256 16 --scene one_sphere
  REINFOCUS_HIP_LIB=$PWD/$lib timeout -k 10 300 python tools/bench_general.py 64 300 100 --scene one_sphere
done; done 2>&1 | tee gpurun_out/r04_ab_occ.txt
