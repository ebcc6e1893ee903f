#!/bin/bash
set -u
OUT=gpurun_out/r03_q; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -m gpu -x -q --durations=5 > $OUT/pytest.log 2>&1; echo "pytest rc=$? $(tail -1 $OUT/pytest.log)"
