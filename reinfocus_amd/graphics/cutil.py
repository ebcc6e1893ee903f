"""Launch-shape helpers with the reference's semantics (reinfocus/graphics/cutil.py:16-104).

The gfx950 kernels fix their own launch geometry (256-thread blocks, one tile of one environment
per block: see rf_abi_render.hip pick_tile_layout), so nothing here sizes a launch.  The functions exist
because `block_shape` is part of FastRenderer's and render()'s signatures: it is normalised with
the reference's rules (limit_block_size) and checked, and `enough_blocks` answers the same
questions the reference's callers and tests ask of it.
"""

import math

MAX_BLOCK_SIZE = 1024  # cutil.py:13 CUDA_MAX_BLOCK_SIZE; also HIP's limit on gfx950


def _is_line(shape):
    return isinstance(shape, int)


def enough_blocks(shape, block_shape):
    """Blocks of block_shape needed to cover shape (cutil.py:16-32): ceil of the elementwise
    quotient, an int for ints, a tuple for tuples."""
    if _is_line(shape):
        return int(math.ceil(shape / block_shape))
    return tuple(int(math.ceil(s / b)) for s, b in zip(shape, block_shape))


def constant_like(n, shape):
    """A shape like `shape` filled with n, never larger than shape (cutil.py:35-48)."""
    if _is_line(shape):
        return min(n, shape)
    return tuple(min(n, s) for s in shape)


def limit_block_size(block_size):
    """Halves the largest side until the block has at most MAX_BLOCK_SIZE threads
    (cutil.py:51-73)."""
    if _is_line(block_size):
        return min(MAX_BLOCK_SIZE, block_size)
    sides = list(block_size)
    while math.prod(sides) > MAX_BLOCK_SIZE:
        sides[sides.index(max(sides))] //= 2
    return tuple(sides)


def launch_shapes(shape, block_shape=None):
    """(blocks_per_grid, threads_per_block) the reference's launcher would use for `shape`
    (cutil.py:76-104): blocks of 16 per side unless given, limited to MAX_BLOCK_SIZE threads."""
    threads = limit_block_size(constant_like(16, shape) if block_shape is None else block_shape)
    return enough_blocks(shape, threads), threads


def check_block_shape(block_shape, dimensions=3):
    """Normalises a caller's block_shape as the reference's launcher would and rejects what numba
    would reject at launch (wrong rank, non-positive sides).  The result does not change the
    gfx950 launch -- frames and RNG streams do not depend on the launch geometry."""
    if _is_line(block_shape):
        block_shape = (block_shape,)
    block_shape = tuple(int(b) for b in block_shape)
    assert len(block_shape) == dimensions, f"block_shape must have {dimensions} sides, not {block_shape}"
    assert all(b > 0 for b in block_shape), f"block_shape sides must be positive, not {block_shape}"
    return limit_block_size(block_shape)
