#!/bin/bash
# round 4, call 7: what the in-wave tails of the general renderer cost (timing builds, wrong frames): the dense pass with
# in-wave loops capped at 0 + 1 / 1 + 2 trips (disc + sphere) beyond the first attempt
export REINFOCUS_GENERAL_DENSE=2
bash tools/ab_general.sh reinfocus_amd/libreinfocus_hip.so tools/lib_cap0.so tools/lib_cap1.so 2>&1 | tee gpurun_out/r04_g.txt
