#!/bin/bash
# Development tool: bench_general.py with several builds of libreinfocus_hip.so (GPU box).
# usage: bash tools/ab_general.sh tools/lib_a.so tools/lib_b.so ...
for lib in "$@"; do
  echo "== $lib"
  REINFOCUS_HIP_LIB=$PWD/$lib timeout -k 10 300 python tools/bench_general.py || exit 1
done
