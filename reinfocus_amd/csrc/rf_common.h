// rf_common.h -- what every kernel header of libreinfocus_hip.so shares.
//
// Data layout in HBM (all owned by rf_ctx):
//   states  uint64x2[n*h*w]   index = e*h*w + y*w + x  (render.py:217) -> a wave's 64
//                             lanes read/write 1 KiB contiguous (global_*_dwordx4)
//   frames  uint8[n][h][w][3] lanes along x; tiles are staged in LDS and leave as coalesced dword stores
//   cam_dyn float[n][9], rect float[n][2]   per-env parameters (block-uniform)
#pragma once

#include <hip/hip_runtime.h>

#include "rf_math.h"

namespace rf {

constexpr int kBlock = 256;

__device__ __forceinline__ bool skip_env(const float *rect, int e)
{
    return __builtin_bit_cast(uint32_t, rect[2 * (size_t)e]) == kSkipEnvBits;
}

// The scene arrays (cameras, shape parameters) are written before the launch (by the host or by an earlier kernel) and never by the kernel that reads them: read
// through the constant address space, a block-uniform address becomes an s_load into scalar registers.  (Through a plain
// pointer the compiler has to assume that the kernel's own stores and atomics may have changed them: it then re-reads
// them after every barrier with one vector load per lane.)
template <class T>
using const_as = const __attribute__((address_space(4))) T;
template <class T>
__device__ __forceinline__ const_as<T> *as_const(const T *p)
{
    return (const_as<T> *)(unsigned long long)p;
}

} // namespace rf
