#!/bin/bash
# round 4, call 19: in-wave disc tails (DW) / per-wave slots (WS) on the instances for frames that are not powers of two, re-measured on the
# fenced kernel (round 3: -1.2 % / -1.3...-2.3 %); bit-identical (parity test on each)
for lib in tools/lib_dw1ws0.so tools/lib_dw1ws1.so tools/lib_dw0ws1.so; do REINFOCUS_HIP_LIB=$PWD/$lib timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -n 1; done
AB_ARGS="--envs-per-gpu 512 --frame 300 --spp 100" REPS=3 bash tools/ab.sh tools/lib_dw0ws0.so tools/lib_dw1ws0.so tools/lib_dw1ws1.so tools/lib_dw0ws1.so 2>&1 | tee gpurun_out/r04_s_300.txt
AB_ARGS="--envs-per-gpu 1024 --frame 384 --spp 16" REPS=2 bash tools/ab.sh tools/lib_dw0ws0.so tools/lib_dw1ws0.so tools/lib_dw1ws1.so tools/lib_dw0ws1.so 2>&1 | tee gpurun_out/r04_s_384.txt
AB_ARGS="--envs-per-gpu 256 --frame 600 --spp 32" REPS=2 bash tools/ab.sh tools/lib_dw0ws0.so tools/lib_dw1ws0.so tools/lib_dw1ws1.so tools/lib_dw0ws1.so 2>&1 | tee gpurun_out/r04_s_600.txt
