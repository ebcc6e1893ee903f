#!/bin/bash
set -u
{ echo "## automatic kernel choice (few blocks: the kernel without cooperative tails): tests/soak_render.py 500 22"; timeout -k 10 900 python tests/soak_render.py 500 22 2>&1 | tail -n 1
echo "## automatic: tools/soak_env.py 200 23"; timeout -k 10 600 python tools/soak_env.py 200 23 2>&1 | tail -n 1; } | tee gpurun_out/r04_ar.txt
