#!/bin/bash
# Development tool: register / LDS / spill figures of the headline render kernel instance for a set of -D flags.
# usage: bash tools/regs.sh "-DRF_ROT=1"   [kernel-mangled-substring]
K=${2:-render_kernel_coop2ILb1ELi1ELi4ELi32E}
cd "$(dirname "$0")/../reinfocus_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt \
  -fno-fast-math -fno-slp-vectorize -mllvm -amdgpu-atomic-optimizer-strategy=None $1 -S --cuda-device-only -o /tmp/regs_$$.s rf_abi.hip -Rpass-analysis=kernel-resource-usage 2>&1 \
  | grep -A10 "Function Name: _ZN2rf19$K" | grep -E "VGPRs:|Spill|Occupancy|LDS Size|ScratchSize" | tr -s ' ' | tr '\n' ';'
echo
rm -f /tmp/regs_$$.s
