// rf_general_chunk.h -- how many environments one launch of the general renderer's listed kernels takes (plain C++: the host
// code of rf_abi_general.hip, and tests/hostsim for tests/test_general_renderer.py).
#pragma once

#include <stdint.h>

namespace rf {

// blocks per environment of render_general_dense_kernel: 16 x 16 tiles (waves of 8 x 8 pixels) unless they pad the frame
// much more than runs of 256 pixels do (frames narrower or lower than a tile); *tiled says which
inline uint64_t dense_blocks_per_env(int h, int w, bool *tiled)
{
    const uint64_t gx = ((uint64_t)h * (uint64_t)w + 255) / 256;
    const uint64_t tiles = (uint64_t)((w + 15) / 16) * (uint64_t)((h + 15) / 16);
    *tiled = tiles * 256 * 100 <= gx * 256 * 115;
    return *tiled ? tiles : gx;
}

// blocks per environment of render_general_one_kernel: tiles of 128 x 6 or of 64 x 12 (narrow), whichever leaves fewer dead
// columns (sets = pixels per thread: rf_coop2.h kSets)
inline uint64_t one_blocks_per_env(int h, int w, int sets, bool *narrow)
{
    *narrow = ((w + 63) / 64) * 64 < ((w + 127) / 128) * 128;
    return *narrow ? (uint64_t)((w + 63) / 64) * (uint64_t)((h + 4 * sets - 1) / (4 * sets))
                   : (uint64_t)((w + 127) / 128) * (uint64_t)((h + 2 * sets - 1) / (2 * sets));
}

// Environments per launch of a kernel with a fix-up list and a one-dimensional grid of blocks_per_env * ne blocks of 256
// threads: pixel indices fit 32 bits (ne * h * w <= 2^32 - 1), and so does the number of launched THREADS -- HIP rejects a
// launch whose gridDim.x * blockDim.x exceeds 2^32 - 1 (hipErrorInvalidConfiguration), and the padded tiles launch up to
// 15 % more threads than the frame has pixels (300 x 300: 361 tiles = 92 416 threads for 90 000 pixels).
inline int general_listed_chunk(uint64_t hw, uint64_t blocks_per_env)
{
    uint64_t chunk = 65535;
    if (0xFFFFFFFFull / hw < chunk)
        chunk = 0xFFFFFFFFull / hw;
    if (0xFFFFFFFFull / (blocks_per_env * 256) < chunk)
        chunk = 0xFFFFFFFFull / (blocks_per_env * 256);
    return (int)(chunk < 1 ? 1 : chunk);
}

} // namespace rf
