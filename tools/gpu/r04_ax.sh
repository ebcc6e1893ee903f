#!/bin/bash
# round 4, last call: long randomised parity soaks on the final tree (library's kernel choice, and three pixels per thread / single-shape kernel forced)
set -u
{ echo "## tests/soak_render.py 1200 51 (automatic kernel choice)"; timeout -k 10 1000 python tests/soak_render.py 1200 51 2>&1 | tail -n 1
echo "## REINFOCUS_RENDER_SETS=3 tests/soak_render.py 1200 52"; REINFOCUS_RENDER_SETS=3 timeout -k 10 1000 python tests/soak_render.py 1200 52 2>&1 | tail -n 1
echo "## tools/soak_env.py 500 53 (automatic)"; timeout -k 10 1000 python tools/soak_env.py 500 53 2>&1 | tail -n 1
echo "## REINFOCUS_RENDER_SETS=3 tools/soak_env.py 500 54"; REINFOCUS_RENDER_SETS=3 timeout -k 10 1000 python tools/soak_env.py 500 54 2>&1 | tail -n 1
echo "## tests/soak_general.py 800 55 (automatic)"; timeout -k 10 1000 python tests/soak_general.py 800 55 2>&1 | tail -n 1
echo "## REINFOCUS_GENERAL_ONE=1 tests/soak_general.py 800 56"; REINFOCUS_GENERAL_ONE=1 timeout -k 10 1000 python tests/soak_general.py 800 56 2>&1 | tail -n 1; } | tee gpurun_out/r04_ax.txt
