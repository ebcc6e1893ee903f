#!/bin/bash
# round 4, call 18: LLVM scheduling strategies for the render kernel (bit-identical builds; parity test on each)
for lib in tools/lib_iterilp.so tools/lib_minreg.so; do REINFOCUS_HIP_LIB=$PWD/$lib timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -n 1; done
REPS=3 bash tools/ab.sh tools/lib_base.so tools/lib_iterilp.so tools/lib_minreg.so 2>&1 | tee gpurun_out/r04_r_head.txt
AB_ARGS="--envs-per-gpu 512 --frame 300 --spp 100" REPS=2 bash tools/ab.sh tools/lib_base.so tools/lib_iterilp.so tools/lib_minreg.so 2>&1 | tee gpurun_out/r04_r_300.txt
