#!/bin/bash
set -u
timeout -k 10 600 python -m pytest tests/test_gpu_general.py tests/test_gpu_notebook_figures.py -x -q -m gpu 2>&1 | tail -n 2
timeout -k 10 300 python tools/bench_general.py 256 256 16 --scene one_rect
timeout -k 10 300 python tools/bench_general.py 64 300 100 --scene one_rect
timeout -k 10 300 python tools/bench_general.py 256 256 16
