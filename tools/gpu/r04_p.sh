#!/bin/bash
# round 4, call 16: delayed waves (RF_TEST_SKEW): the shipped ordering holds, the round-3 form does not
set -u
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "delayed or overflow" 2>&1 | tail -n 3
timeout -k 10 300 python tools/gpu/r04_skew.py 2>&1 | tee gpurun_out/r04_skew.txt
