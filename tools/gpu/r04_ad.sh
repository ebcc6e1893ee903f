#!/bin/bash
# round 4, call 35: the general renderer's sample / bounce loops as one loop over intersection events (lanes start their next sample while neighbours bounce)
set -u -o pipefail
timeout -k 10 900 python -m pytest tests/test_gpu_general.py tests/test_gpu_notebook_figures.py -x -q -m gpu 2>&1 | tail -n 3 || exit 1
for rep in 1 2; do for lib in tools/lib_nested.so reinfocus_amd/libreinfocus_hip.so; do
  echo "== $lib"
  REINFOCUS_GENERAL_ONE=0 REINFOCUS_HIP_LIB=$PWD/$lib timeout -k 10 300 python tools/bench_general.py 256 256 16
  REINFOCUS_GENERAL_ONE=0 REINFOCUS_HIP_LIB=$PWD/$lib timeout -k 10 300 python tools/bench_general.py 64 300 100
done; done 2>&1 | tee gpurun_out/r04_ad.txt
