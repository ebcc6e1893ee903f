#!/bin/bash
# round 4, call 57: where the single-shape kernel overtakes the literal one
set -u
for cfg in "24 300 100" "32 300 100" "48 300 100" "16 256 16" "32 256 16" "64 256 16" "8 512 16"; do for one in 1 0; do
  echo "== $cfg REINFOCUS_GENERAL_ONE=$one"
  REINFOCUS_GENERAL_ONE=$one timeout -k 10 300 python tools/bench_general.py $cfg --scene one_rect
  REINFOCUS_GENERAL_ONE=$one timeout -k 10 300 python tools/bench_general.py $cfg --scene one_sphere
done; done 2>&1 | tee gpurun_out/r04_at.txt
