#!/bin/bash
# round 4, call 56: the general renderer's single-shape kernel on very small launches (the notebooks render 1-2 environments) against the literal kernel
set -u
for n in 1 2 4 8 16; do for one in 1 0; do
  echo "== n=$n REINFOCUS_GENERAL_ONE=$one"
  REINFOCUS_GENERAL_ONE=$one timeout -k 10 300 python tools/bench_general.py $n 300 100 --scene one_rect
  REINFOCUS_GENERAL_ONE=$one timeout -k 10 300 python tools/bench_general.py $n 300 100 --scene one_sphere
done; done 2>&1 | tee gpurun_out/r04_as.txt
