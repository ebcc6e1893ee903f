#!/bin/bash
# round 3, GPU call O: PC sampling of the render kernel (beta feature: short timeouts, small run first)
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03_o; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 120 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-unit time --pc-sampling-method host_trap --pc-sampling-interval 1 \
   --output-format csv -d $OUT/pcs_small -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pmc --envs-per-gpu 256 --frame 128 --spp 4 > $OUT/pcs_small.log 2>&1
echo "small host_trap rc=$?"; tail -3 $OUT/pcs_small.log | cut -c1-300; ls -la $OUT/pcs_small/* 2>/dev/null | head
