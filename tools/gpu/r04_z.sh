#!/bin/bash
# round 4, call 26: full GPU suite + env soak on the fused env step
set -u -o pipefail
timeout -k 10 1000 python -m pytest tests -x -q -m gpu 2>&1 | tail -n 4 || exit 1
timeout -k 10 600 python tools/soak_env.py 150 31 2>&1 | tail -n 1
