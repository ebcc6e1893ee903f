#!/bin/bash
# round 4, call 22: the cooperative single-rectangle general kernel: parity, then throughput by register budget
set -u
OUT=gpurun_out/r04_v; mkdir -p $OUT; rm -f $OUT/*
timeout -k 10 600 python -m pytest tests/test_gpu_general.py tests/test_gpu_notebook_figures.py tests/test_abi_library.py -x -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -n 5 $OUT/pytest.log
echo "== literal"; REINFOCUS_GENERAL_ONE=0 timeout -k 10 300 python tools/bench_general.py 256 256 16 --scene one_rect 2>&1 | tee $OUT/literal.txt
for lib in reinfocus_amd/libreinfocus_hip.so tools/lib_rocc7.so tools/lib_rocc5.so; do echo "== $lib"; REINFOCUS_HIP_LIB=$PWD/$lib timeout -k 10 300 python tools/bench_general.py 256 256 16 --scene one_rect; REINFOCUS_HIP_LIB=$PWD/$lib timeout -k 10 300 python tools/bench_general.py 64 300 100 --scene one_rect; done 2>&1 | tee $OUT/ab.txt
