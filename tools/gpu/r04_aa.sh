#!/bin/bash
# round 4, call 30: single-sphere worlds on the cooperative single-shape kernel
set -u -o pipefail
timeout -k 10 900 python -m pytest tests/test_gpu_general.py tests/test_gpu_notebook_figures.py -x -q -m gpu 2>&1 | tail -n 3 || exit 1
mkdir -p gpurun_out/r04_aa
for rep in 1 2; do
timeout -k 10 300 python tools/bench_general.py 256 256 16 2>&1 | tee -a gpurun_out/r04_aa/bench_general.txt
REINFOCUS_GENERAL_RECT=0 timeout -k 10 300 python tools/bench_general.py 256 256 16 --scene one_sphere 2>&1 | tee -a gpurun_out/r04_aa/bench_general.txt
done
timeout -k 10 300 python tools/bench_general.py 64 300 100 2>&1 | tee -a gpurun_out/r04_aa/bench_general.txt
timeout -k 10 600 python tests/soak_general.py 300 23 2>&1 | tail -n 1
