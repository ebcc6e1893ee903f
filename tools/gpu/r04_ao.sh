#!/bin/bash
# round 4, call 51: randomised parity soaks with the library's own kernel choice (few blocks: one pixel per thread) and with three pixels per thread forced
set -u
{ echo "## automatic kernel choice: tests/soak_render.py 500 17"; timeout -k 10 900 python tests/soak_render.py 500 17 2>&1 | tail -n 1
echo "## REINFOCUS_RENDER_SETS=3: tests/soak_render.py 500 18"; REINFOCUS_RENDER_SETS=3 timeout -k 10 900 python tests/soak_render.py 500 18 2>&1 | tail -n 1
echo "## automatic: tools/soak_env.py 200 19"; timeout -k 10 600 python tools/soak_env.py 200 19 2>&1 | tail -n 1
echo "## REINFOCUS_RENDER_SETS=3: tools/soak_env.py 200 20"; REINFOCUS_RENDER_SETS=3 timeout -k 10 600 python tools/soak_env.py 200 20 2>&1 | tail -n 1
echo "## tests/soak_general.py 400 21"; timeout -k 10 600 python tests/soak_general.py 400 21 2>&1 | tail -n 1; } | tee gpurun_out/r04_ao.txt
