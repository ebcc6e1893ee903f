// probe_general.h -- TEST INFRASTRUCTURE: the float64 library calls and the two helpers of the
// general renderer (reinfocus_amd/csrc/rf_general.h) that depend on them, as plain functions
// compiled once for the GPU (gpucheck.hip) and once for the host (tests/hostsim): comparing the
// two on the same operands shows which operation of the device math library differs from glibc.
#pragma once

#include "../../reinfocus_amd/csrc/rf_general.h"

namespace probe {

// op: 0 sqrt(a)  1 a / b  2 atan2(a, b)  3 acos(a)  4 sin(a)  5 (atan2(a, b) + pi) / pi  6 acos(a) / pi
RF_HD double f64_op(int op, double a, double b)
{
    switch (op) {
    case 0: return sqrt(a);
    case 1: return a / b;
    case 2: return atan2(a, b);
    case 3: return acos(a);
    case 4: return sin(a);
    case 5: return (atan2(a, b) + rf::kPi) / rf::kPi;
    default: return acos(a) / rf::kPi;
    }
}

// sphere.uv as sphere_hit computes it from a surface normal
RF_HD void sphere_uv(const float n[3], float uv[2])
{
    uv[0] = (float)((atan2(-(double)n[2], (double)n[0]) + rf::kPi) / rf::kPi);
    uv[1] = (float)(acos(-(double)n[1]) / rf::kPi);
}

// the checker colour of a sphere hit: the float32 fast path with its fallback, and the reference's
// float64 expressions alone
RF_HD int sphere_red_fast(const float n[3], float fu, float fv) { return rf::sphere_red(n, fu, fv) ? 1 : 0; }
RF_HD void sphere_uv_fast(const float n[3], float uv[2]) { rf::sphere_uv_approx(n, uv[0], uv[1]); }
RF_HD int sphere_red_exact(const float n[3], float fu, float fv) { return rf::sphere_red_exact(n, fu, fv) ? 1 : 0; }

} // namespace probe
