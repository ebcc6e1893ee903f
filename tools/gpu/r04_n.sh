#!/bin/bash
# round 4, call 14: the cooperative multi-set general renderer: parity, then throughput by register budget / sets per thread
set -u
OUT=gpurun_out/r04_n; mkdir -p $OUT; rm -f $OUT/*
timeout -k 10 600 python -m pytest tests/test_gpu_general.py tests/test_gpu_notebook_figures.py tests/test_abi_library.py -x -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -n 5 $OUT/pytest.log
echo "== literal"; REINFOCUS_GENERAL_COOP=0 timeout -k 10 300 python tools/bench_general.py 256 256 16 2>&1 | tee $OUT/literal.txt
bash tools/ab_general.sh reinfocus_amd/libreinfocus_hip.so tools/lib_gocc4.so tools/lib_gocc6.so tools/lib_g3s.so 2>&1 | tee $OUT/ab.txt
