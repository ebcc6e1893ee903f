#!/bin/bash
# round 4, call 38: in-wave disc tails (DW) / per-wave list slots (WS) on the frames that are not powers of two, now that these instances do not spill
set -u
{ AB_ARGS="--envs-per-gpu 512 --frame 300 --spp 100" REPS=3 bash tools/ab.sh tools/lib_base.so tools/lib_dwws.so tools/lib_ws.so tools/lib_dw.so
AB_ARGS="--envs-per-gpu 256 --frame 600 --spp 16" REPS=2 bash tools/ab.sh tools/lib_base.so tools/lib_dwws.so tools/lib_ws.so tools/lib_dw.so
AB_ARGS="--envs-per-gpu 2048 --frame 100 --spp 16" REPS=2 bash tools/ab.sh tools/lib_base.so tools/lib_dwws.so tools/lib_ws.so tools/lib_dw.so
AB_ARGS="--envs-per-gpu 1024 --frame 384 --spp 16" REPS=2 bash tools/ab.sh tools/lib_base.so tools/lib_dwws.so tools/lib_ws.so tools/lib_dw.so; } 2>&1 | tee gpurun_out/r04_af.txt
