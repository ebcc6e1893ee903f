"""CPU-side checks of the render kernel's arithmetic.

tests/hostsim compiles reinfocus_amd/csrc/rf_math.h -- the exact text the gfx950 kernel
inlines -- for the host.  These tests prove on the CPU that each exact rewrite used by the
kernel (32-bit-word RNG, one-fma uniform conversion, pow2 pixel coordinates, checker sign
table, AXIS camera specialisation, GF(2) jump tables) equals the literal form, so GPU time
is only spent confirming the device code generation.  Test infrastructure; the product
never loads this library."""

import ctypes
import os
import subprocess

import numpy as np
import pytest

from tests import helpers

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def hs():
    lib = ctypes.CDLL(helpers.built("tests/hostsim", "libhostsim.so"))
    p = ctypes.c_void_p
    lib.hs_check_uniform.restype = ctypes.c_long
    lib.hs_check_uniform.argtypes = [p, ctypes.c_long]
    lib.hs_check_pixel_coord.restype = ctypes.c_long
    lib.hs_check_pixel_coord.argtypes = [p, ctypes.c_long, ctypes.c_int]
    lib.hs_check_pixel_coord_div.restype = ctypes.c_long
    lib.hs_check_pixel_coord_div.argtypes = [p, ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    lib.hs_check_lens.restype = ctypes.c_long
    lib.hs_check_lens.argtypes = [ctypes.c_double]
    lib.hs_check_checker.restype = ctypes.c_long
    lib.hs_check_checker.argtypes = [p, p, ctypes.c_long]
    lib.hs_render.argtypes = [p] + [ctypes.c_int] * 4 + [p] * 5 + [ctypes.c_double, p, ctypes.c_int]
    lib.hs_state_at.argtypes = [ctypes.c_uint64, ctypes.c_uint64, p]
    lib.hs_check_div.restype = ctypes.c_long
    lib.hs_check_div.argtypes = [p, ctypes.c_long, p, ctypes.c_long]
    lib.hs_check_sky.restype = ctypes.c_long
    lib.hs_check_accept.restype = ctypes.c_long
    lib.hs_check_accept.argtypes = [p, ctypes.c_long, ctypes.c_int, p]
    lib.hs_variance_from_sums.restype = ctypes.c_double
    lib.hs_variance_from_sums.argtypes = [ctypes.c_uint64] * 3
    return lib


def _edge_words():
    words = []
    mask = (1 << 64) - 1
    for sh in range(64):
        for pat in (1, 3, 0x7FF, 0x800, 0xFFFFFF, 0x1000001, 0x1FFFFFF, 0x3000000, (1 << 53) - 1, mask,
                    0x8000008000000000, 0x1FF, 0x200, 0x201):
            for delta in (0, 1, -1):
                words.append(((pat << sh) + delta) & mask)
    # ties: 25 significant bits, with and without sticky bits below
    for top in range(11, 64):
        for k in (24, 25, 26):
            if top - k >= 0:
                for extra in (0, 1, 1 << max(0, top - k - 1), (1 << max(0, top - k)) - 1):
                    words.append(((1 << top) | (1 << (top - k)) | extra) & mask)
                    words.append(((1 << top) | (((1 << 24) - 1) << max(0, top - 23)) | (1 << (top - k)) | extra) & mask)
    # small high words (the rare general path) with every low-bit pattern class
    for hi in (0, 1, 2, 255, 256, 511, 512, 513, 1023):
        for lo in (0, 0x800, 0xF800, 0x10000, 0xFFFFFFFF, 0x80000000, 0x7FF):
            words.append((hi << 32) | lo)
    return np.array(words, dtype=np.uint64)


def test_uniform_conversions_equal_literal(hs):
    rng = np.random.default_rng(1)
    words = np.concatenate([
        rng.integers(0, 2**64, size=4_000_000, dtype=np.uint64),
        _edge_words(),
        np.arange(0, 1 << 16, dtype=np.uint64),
        np.arange(0, 1 << 16, dtype=np.uint64) << np.uint64(48),
        np.arange(0, 1 << 16, dtype=np.uint64) << np.uint64(24),
        # random low words under small high words
        (rng.integers(0, 1024, size=500_000, dtype=np.uint64) << np.uint64(32))
        | rng.integers(0, 2**32, size=500_000, dtype=np.uint64),
    ])
    assert hs.hs_check_uniform(words.ctypes.data, len(words)) == 0


def test_pixel_coord_pow2_equals_literal(hs):
    rng = np.random.default_rng(2)
    xis = np.concatenate([
        rng.random(5000, dtype=np.float32),
        np.array([0, 1, 2.0**-53, 2.0**-30, 2.0**-24, 2.0**-25, 1 - 2.0**-24, 0.5, 2.0**-41, 2.0**-42, 2.0**-29],
                 dtype=np.float32),
        (rng.random(5000) * 2.0 ** -rng.integers(0, 50, 5000)).astype(np.float32),
    ])
    for w in (1, 2, 64, 128, 256, 512, 1024, 4096):
        assert hs.hs_check_pixel_coord(xis.ctypes.data, len(xis), w) == 0


def test_pixel_coord_division_free_form(hs):
    """s = float32((x + xi) / w) for frame sizes that are not powers of two: the 3-op Markstein
    quotient with the host's RN64(1 / w) equals the IEEE division for every w in [1, 4096]
    (x strided to ~64 columns per size, 4100 jitters each: 1.1e9 quotients) and for every x of
    the reference's own sizes 300 and 600."""
    rng = np.random.default_rng(17)
    xis = np.concatenate([
        rng.random(2000, dtype=np.float32),
        np.array([0, 1, 2.0**-53, 2.0**-30, 2.0**-24, 2.0**-25, 1 - 2.0**-24, 0.5, 2.0**-41, 2.0**-42, 2.0**-29],
                 dtype=np.float32),
        (rng.random(2089) * 2.0 ** -rng.integers(0, 50, 2089)).astype(np.float32),
    ])
    assert hs.hs_check_pixel_coord_div(xis.ctypes.data, len(xis), 1, 4096, 0) == 0
    dense = rng.random(100_000, dtype=np.float32)
    for w in (300, 600, 100, 720, 1080):
        assert hs.hs_check_pixel_coord_div(dense.ctypes.data, len(dense), w, w, 1) == 0


def test_lens_offset_float32_form(hs):
    """offset = float32(float64(p) * lens_radius): for the reference's aperture 0.1 (radius
    numpy.divide(0.1, 2.0) = 0.05) the float32 fma form is exact for all 25 165 825 possible disc
    coordinates; radii for which it is not exist, and the C ABI then keeps the float64 form (the
    same exhaustive check runs in rf_abi_ctx.hip lens_split)."""
    assert hs.hs_check_lens(float(np.divide(0.1, 2.0))) == 0
    assert hs.hs_check_lens(0.0625) == 0  # a power of two scales exactly
    # about one radius in twelve has a few coordinates that round differently, e.g.:
    assert hs.hs_check_lens(0.6243510725689605) == 1
    assert hs.hs_check_lens(0.46456785704581477) == 5


def test_checker_sign_equals_libm_sin(hs):
    rng = np.random.default_rng(3)
    special = [np.float32(k / 32) for k in range(33)]
    for k in range(33):
        special.append(np.nextafter(np.float32(k / 32), np.float32(2)))
        if k:
            special.append(np.nextafter(np.float32(k / 32), np.float32(-1)))
    special += [np.float32(1e-45), np.float32(1e-38), np.float32(1e-30), np.float32(1e-10)]
    special = np.array(special, dtype=np.float32)
    us = np.concatenate([rng.random(500_000, dtype=np.float32), special])
    vs = us[rng.permutation(len(us))].copy()
    assert hs.hs_check_checker(us.ctypes.data, vs.ctypes.data, len(us)) == 0
    a, b = np.meshgrid(special, special)
    a, b = np.ascontiguousarray(a.ravel()), np.ascontiguousarray(b.ravel())
    assert hs.hs_check_checker(a.ctypes.data, b.ctypes.data, len(a)) == 0


def test_div_by_const_equals_ieee_division(hs):
    """uv = (p + half) / (half + half) through the 3-op Markstein form: 20000 divisors
    (rectangle sides for targets in [0.001, 1e6], r_size 20 and 30, plus random floats)
    x 50000 numerators each = 1e9 quotients, plus the k/32 checker boundaries."""
    import math

    rng = np.random.default_rng(21)
    targets = np.concatenate([rng.uniform(5, 10, 8000), 10 ** rng.uniform(-3, 6, 8000)]).astype(np.float32)
    halves = [(targets.astype(np.float64) * math.tan(math.radians(r / 2))).astype(np.float32) for r in (20, 30)]
    dens = np.concatenate([h + h for h in halves] + [rng.uniform(0.01, 100, 4000).astype(np.float32)])[:20000]
    fracs = np.concatenate([rng.random(49000, dtype=np.float32), np.linspace(0, 1, 1000, dtype=np.float32)])
    dens, fracs = np.ascontiguousarray(dens), np.ascontiguousarray(fracs)
    assert hs.hs_check_div(dens.ctypes.data, len(dens), fracs.ctypes.data, len(fracs)) == 0


def test_sky_f32_equals_literal(hs):
    """Every float32 ud.y in [-1 - 8 ulp, 1 + 8 ulp] (2.1e9 values): the four f32 fma forms
    of the sky colour equal the reference's float64 chain."""
    assert hs.hs_check_sky() == 0


def _words_for(points):
    """64-bit draw words whose uniform value u satisfies 2u - 1 ~= point."""
    u = np.clip((np.asarray(points, dtype=np.float64) + 1.0) * 0.5, 0.0, 1.0 - 2.0**-53)
    return (u * 2.0**64).astype(np.uint64)


@pytest.mark.parametrize("dims", [2, 3])
def test_rejection_decision_equals_literal(hs, dims):
    """Random candidates plus candidates placed on and around the unit circle / sphere
    (radius 1 +- up to 3e-5, both sides of the approximate test's band)."""
    rng = np.random.default_rng(10 + dims)
    n = 400_000
    rand = rng.integers(0, 2**64, size=(n, dims), dtype=np.uint64)
    v = rng.normal(size=(n, dims))
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    radius = 1.0 + rng.uniform(-3e-5, 3e-5, size=(n, 1)) * rng.integers(0, 2, size=(n, 1))
    near = _words_for(v * radius)
    near ^= rng.integers(0, 2**20, size=near.shape, dtype=np.uint64)  # scramble the low bits
    axis = np.zeros((dims * 2, dims))
    for k in range(dims):
        axis[2 * k, k], axis[2 * k + 1, k] = 1.0, -1.0
    words = np.ascontiguousarray(np.concatenate([rand, near, _words_for(axis)]))
    in_band = ctypes.c_long(0)
    bad = hs.hs_check_accept(words.ctypes.data, len(words), dims, ctypes.byref(in_band))
    assert bad == 0
    assert in_band.value > 10_000  # the band path really ran


def test_jump_tables_equal_sequential_jumps(hs, oracle):
    ref = oracle.seed_states(1100, 0)
    out = np.zeros(2, dtype=np.uint64)
    for i in (0, 1, 2, 63, 64, 65, 127, 255, 256, 1023, 1099):
        assert hs.hs_state_at(0, i, out.ctypes.data) == 0
        assert np.array_equal(out, ref[i])
    ref7 = oracle.seed_states(40, 7)
    assert hs.hs_state_at(7, 39, out.ctypes.data) == 0 and np.array_equal(out, ref7[39])


def _hs_render(hs, scene, n, h, w, spp, states, mode):
    dyn, rect, origin, u, v, lens = scene
    frames = np.zeros((n, h, w, 3), dtype=np.uint8)
    origin, u, v = (np.ascontiguousarray(x, dtype=np.float32) for x in (origin, u, v))
    hs.hs_render(frames.ctypes.data, n, h, w, spp, dyn.ctypes.data, rect.ctypes.data, origin.ctypes.data,
                 u.ctypes.data, v.ctypes.data, lens, states.ctypes.data, mode)
    return frames


@pytest.mark.parametrize("n,h,w,spp", [(6, 32, 32, 8), (3, 30, 30, 7), (4, 64, 16, 3), (2, 20, 50, 5)])
def test_kernel_arithmetic_equals_oracle(hs, oracle, n, h, w, spp):
    """mode bit 0 = AXIS specialisation, bit 1 = POW2 specialisation."""
    rng = np.random.default_rng(n * 100 + h)
    targets, focus = helpers.random_scene(rng, n)
    scene = helpers.pack_scene(targets, focus)
    st0 = oracle.seed_states(n * h * w, 0)
    st = st0.copy()
    want = oracle.render(scene[0], scene[1], h, w, spp, st)
    pow2 = (h & (h - 1)) == 0 and (w & (w - 1)) == 0
    for mode in ([0, 1, 2, 3] if pow2 else [0, 1]):
        s2 = st0.copy()
        got = _hs_render(hs, scene, n, h, w, spp, s2, mode)
        assert np.array_equal(got, want), f"mode {mode}"
        assert np.array_equal(s2, st), f"mode {mode}"


def test_kernel_arithmetic_extreme_scenes(hs, oracle):
    targets = np.array([1.0, 40.0, 0.0005, 2.0e6, 10.0, 5.0], dtype=np.float32)
    focus = np.array([40.0, 1.0, 10.0, 10.0, 0.01, 1.0e4], dtype=np.float32)
    n, h, w, spp = len(targets), 16, 16, 4
    scene = helpers.pack_scene(targets, focus)
    st0 = oracle.seed_states(n * h * w, 0)
    st = st0.copy()
    want = oracle.render(scene[0], scene[1], h, w, spp, st)
    for mode in (0, 1, 2, 3):
        s2 = st0.copy()
        assert np.array_equal(_hs_render(hs, scene, n, h, w, spp, s2, mode), want)
        assert np.array_equal(s2, st)


def test_variance_from_sums_is_exact_for_any_pixel_count(hs):
    """focus_finalize / env_variance (rf_math.h variance_from_sums): (N S2 - S1^2) / N^2 with a 128-bit numerator.  Up to
    round 5 the numerator was cut to 64 bits: wrong from 3.4e7 pixels on (vision.py:25 scores any image)."""
    from fractions import Fraction

    rng = np.random.default_rng(11)
    for npix in (1, 4, 90000, 65536, 600 * 600, 4096 * 4096, 1 << 25, (1 << 26) + 12345, 1 << 31, 40000 * 40000):
        for _ in range(50):
            # a population of values in [0, 255]: k pixels of value a, the rest of value b (and all-equal frames)
            a, b = (int(v) for v in rng.integers(0, 256, size=2))
            k = int(rng.integers(0, npix + 1))
            s1 = k * a + (npix - k) * b
            s2 = k * a * a + (npix - k) * b * b
            want = Fraction(npix * s2 - s1 * s1, npix * npix)
            got = hs.hs_variance_from_sums(npix, s1, s2)
            assert got == pytest.approx(float(want), rel=4e-16, abs=0.0), (npix, a, b, k)
        assert hs.hs_variance_from_sums(npix, 255 * npix, 255 * 255 * npix) == 0.0
    # the largest numerator a 2^26-pixel frame can produce (half 0, half 255) does not fit 64 bits
    n = 1 << 26
    assert n * (n // 2 * 255 * 255) - (n // 2 * 255) ** 2 >= 1 << 64
    assert hs.hs_variance_from_sums(n, n // 2 * 255, n // 2 * 255 * 255) == 127.5 ** 2
