#!/bin/bash
# Development tool: bench_general.py with several builds of libreinfocus_hip.so (GPU box).
# usage: bash tools/ab_general.sh tools/lib_a.so tools/lib_b.so ...
cp reinfocus_amd/libreinfocus_hip.so /tmp/lib_orig.so
for lib in "$@"; do
  cp "$lib" reinfocus_amd/libreinfocus_hip.so
  echo "== $lib"
  timeout -k 10 300 python tools/bench_general.py
done
cp /tmp/lib_orig.so reinfocus_amd/libreinfocus_hip.so
