#!/bin/bash
# round 4, call 13: round-1 survivors take their round-2 slots with one LDS atomic per worker wave (ballot ranks) instead of one per lane
REPS=3 bash tools/ab.sh tools/lib_base.so tools/lib_r1w.so 2>&1 | tee gpurun_out/r04_m_head.txt
AB_ARGS="--envs-per-gpu 512 --frame 300 --spp 100" REPS=2 bash tools/ab.sh tools/lib_base.so tools/lib_r1w.so 2>&1 | tee gpurun_out/r04_m_300.txt
AB_ARGS="--envs-per-gpu 128 --frame 512 --spp 64" REPS=2 bash tools/ab.sh tools/lib_base.so tools/lib_r1w.so 2>&1 | tee gpurun_out/r04_m_512.txt
