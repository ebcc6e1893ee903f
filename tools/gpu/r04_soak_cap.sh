#!/bin/bash
# randomised parity soaks with the overflow test build (32-entry list, 8 disc slots per wave) as the library under test
set -u
OUT=gpurun_out/r04_soak_cap; mkdir -p $OUT
REINFOCUS_HIP_LIB=$PWD/tests/gpucheck/libreinfocus_cap32.so timeout -k 10 900 python tests/soak_render.py 400 21 > $OUT/soak_render_cap32.log 2>&1; echo "render soak (cap32 build) rc=$? $(tail -1 $OUT/soak_render_cap32.log)"
REINFOCUS_HIP_LIB=$PWD/tests/gpucheck/libreinfocus_cap32.so timeout -k 10 600 python tools/soak_env.py 150 21 > $OUT/soak_env_cap32.log 2>&1; echo "env soak (cap32 build) rc=$? $(tail -1 $OUT/soak_env_cap32.log)"
