# Code-generation flags of every gfx950 build in this repository (the product library, its test builds, tests/gpucheck).
# -ffp-contract=off: the parity contract forbids FMA contraction (HIP's default is fast contraction); explicit fma()
#  calls in rf_math.h are proven-equal rewrites.
# -fhip-fp32-correctly-rounded-divide-sqrt: f32 '/' and sqrtf must be IEEE-exact.
# -fno-slp-vectorize: the SLP vectoriser pairs float operations into v_pk_mul/fma/add_f32, which gfx950 issues on its
#  slow VALU path (4.3 cycles for two operations that cost ~1 cycle each as plain v_mul/fma/add_f32 next to slow-path
#  work: tools/ubench/pairbench); +3.2 % measured.
# -mllvm -amdgpu-atomic-optimizer-strategy=None: rf_coop2.h parks stragglers with one LDS atomic per lane on purpose;
#  the optimizer would rewrite that as ballot + mbcnt + one atomic per wave.
ARCH ?= gfx950
RF_CODEGEN_FLAGS = -O3 -std=c++17 --offload-arch=$(ARCH) -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt \
                   -fno-fast-math -fno-slp-vectorize -mllvm -amdgpu-atomic-optimizer-strategy=None
