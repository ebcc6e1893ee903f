"""GPU parity: the HIP path (through the C ABI) against the CPU oracle and the
committed numpy-1.26 golden vectors.  Bit-exact for frames and RNG states; focus
values to 1e-9 relative (integer-exact sums vs numpy's pairwise float64 variance),
far inside the 1e-4 absolute tolerance BASELINE.json states."""

import os

import numpy as np
import pytest

from tests import helpers

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def native():
    from reinfocus_amd import _native

    assert _native.device_count() >= 1, "no GPU visible: the HIP path cannot run"
    return _native


@pytest.fixture()
def ctx(native, kernel_choice):
    """(every test that renders through this context runs with the benchmarked kernels forced and with the library's
    own choice: conftest.kernel_choice)"""
    c = native.Context(0)
    yield c
    c.close()


# --- seeding -------------------------------------------------------------------------------


@pytest.mark.parametrize("n,first", [(1, 0), (63, 0), (64, 0), (5000, 0), (3000, 123457)])
def test_seed_matches_sequential_jumps(ctx, oracle, n, first):
    """rf_seed's GF(2) jump-ahead == numba's sequential host seeding (random.py:8-18)."""
    ctx.seed(n, 0, first)
    want = oracle.seed_states(first + n, 0)[first:]
    assert np.array_equal(ctx.get_states(), want)


def test_the_suite_runs_on_poisoned_allocations(native):
    """tests/conftest.py starts the session with REINFOCUS_POISON_ALLOC: every buffer the library allocates holds 0xA5 bytes
    before its first use, so a kernel that reads a list nobody cleared or a sum nobody zeroed cannot pass by the luck of a
    fresh page (the same library without the variable: smoke(), bench.py)."""
    assert native.allocations_poisoned()


def test_seed_far_offset(ctx):
    """State indices beyond what the sequential oracle can reach in seconds (the 8-GPU
    shard offsets): compare with the host GF(2) tables, which tests/test_host_logic.py
    pins against sequential jumps."""
    import ctypes

    hs = ctypes.CDLL(os.path.join(os.path.dirname(__file__), "hostsim", "libhostsim.so"))
    hs.hs_state_at.argtypes = [ctypes.c_uint64, ctypes.c_uint64, ctypes.c_void_p]
    n, first = 2049, 2**31 + 5
    ctx.seed(n, 0, first)
    got = ctx.get_states()
    out = np.zeros(2, dtype=np.uint64)
    for i in (0, 1, 63, 64, 65, 2047, 2048):
        assert hs.hs_state_at(0, first + i, out.ctypes.data) == 0
        assert np.array_equal(got[i], out)


def test_seed_other_seed(ctx, oracle):
    ctx.seed(777, 7, 0)
    assert np.array_equal(ctx.get_states(), oracle.seed_states(777, 7))


# --- render: golden vectors ------------------------------------------------------------------


@pytest.mark.parametrize("name", ["render_pow2", "render_npot", "render_rsize30", "render_cfg1", "render_mid"])
def test_render_golden(native, golden_dir, name, kernel_choice):
    """FastRenderer (HIP) reproduces the numpy-1.26 restatement bit for bit."""
    from reinfocus_amd.graphics import render

    g = np.load(os.path.join(golden_dir, name + ".npz"))
    h, spp = int(g["h"]), int(g["spp"])
    r = render.FastRenderer(samples_per_pixel=spp, r_size=float(g["r_size"]), device=0)
    r.update_targets(g["targets"])
    r.update_focus_planes(g["focus"])
    for p in range(int(g["passes"])):
        frames = np.asarray(r.render(h))
        assert frames.shape == g["frames%d" % p].shape and frames.dtype == np.uint8
        assert np.array_equal(frames, g["frames%d" % p])
    n = len(g["targets"])
    assert np.array_equal(r._ctx.get_states(0, n * h * h), g["states_after"])


# --- render: oracle on seeded random scenes ---------------------------------------------------


@pytest.mark.parametrize(
    "n,h,w,spp",
    [
        (8, 64, 64, 4),      # AXIS + POW2
        (3, 100, 100, 3),    # AXIS, non power of two (f64 s/t)
        (5, 33, 35, 2),      # h*w odd: byte-store epilogue
        (2, 128, 32, 5),     # h != w, both powers of two
        (1, 300, 300, 1),    # reference default frame height
        (256, 16, 16, 2),    # many small envs
        (3, 1, 1, 7),        # a single pixel
        (2, 7, 130, 3),      # just over one tile wide, one partial tile row
        (2, 6, 128, 2),      # exactly one tile per env
        (1, 13, 129, 2),     # partial tiles in both directions
    ],
)
def test_render_matches_oracle(ctx, oracle, n, h, w, spp):
    rng = np.random.default_rng(n * 1000 + h)
    targets, focus = helpers.random_scene(rng, n)
    dyn, rect, origin, u, v, lens = helpers.pack_scene(targets, focus)
    st = oracle.seed_states(n * h * w, 0)
    want = oracle.render(dyn, rect, h, w, spp, st, n_threads=8)
    ctx.seed(n * h * w, 0, 0)
    ctx.set_scene(dyn, rect, origin, u, v, lens)
    got = ctx.render(n, h, w, spp, to_host=True)
    assert np.array_equal(got, want)
    assert np.array_equal(ctx.get_states(0, n * h * w), st)
    # second pass continues the same streams
    want2 = oracle.render(dyn, rect, h, w, spp, st, n_threads=8)
    got2 = ctx.render(n, h, w, spp, to_host=True)
    assert np.array_equal(got2, want2)


@pytest.mark.parametrize("n,h,w,spp", [(3, 37, 132, 3), (2, 100, 164, 4), (2, 300, 300, 2), (3, 70, 176, 3), (2, 45, 165, 3),
                                       (3, 33, 148, 5), (1, 16, 600, 2), (2, 24, 232, 3), (2, 21, 179, 2), (2, 20, 100, 2)])
def test_strip_tiles_match_oracle(oracle, tmp_path, n, h, w, spp):
    """Frames wider than 128 whose width leaves 1 .. 48 columns beyond a multiple of 64 are rendered by
    render_kernel_coop2_strip: 128 x 6 or 64 x 12 tiles on the left, 48 x 16 tiles (three pixels per thread side by side)
    for the remainder, in one launch -- remainders of 4, 20, 24 and 40 (main part not a multiple of 128), 36, 37 (odd
    width: byte stores), 44, 48; 51 columns (> 48) and 100-pixel frames keep one tile shape; heights that fill neither
    tile shape's rows.  Frames and RNG states equal the oracle's, and the same frames come out of one tile shape."""
    rng = np.random.default_rng(n * 1000 + w)
    d = helpers.pack_scene(*helpers.random_scene(rng, n))
    states = oracle.seed_states(n * h * w, 0)
    want = oracle.render(d[0], d[1], h, w, spp, states, n_threads=8)  # (advances `states` in place)
    for overrides in ({}, {"REINFOCUS_RENDER_STRIP": "0"}):
        frames, final, kernel = _render_in_child(tmp_path, d, n, h, spp, overrides, w=w, want_kernel=True)
        assert ("strip" in kernel) == (not overrides and w > 128 and w % 64 <= 48), kernel
        assert ("4>" in kernel) == ("strip" in kernel and (w - w % 64) % 128 == 0), kernel
        assert np.array_equal(frames, want), (overrides, kernel)
        assert np.array_equal(final, states), (overrides, kernel)


@pytest.mark.parametrize("h,w", [(32, 32), (24, 40)])
def test_render_general_camera_matches_oracle(ctx, oracle, h, w):
    """A camera frame that is not the canonical one takes the general kernel."""
    from reinfocus_amd.graphics import camera, world

    n, spp = 3, 4
    cams = camera.FastCameras(look_from=(0.5, -0.25, 1.0), look_at=(0.3, 0.2, -9.0), up=(0.1, 1.0, 0.05),
                              aperture=0.3, vfov=35, aspect_ratio=1.5)
    cams.update([6.0, 8.5, 9.75])
    worlds = world.FastWorlds(25)
    worlds.update([7.0, 8.5, 6.0])
    dyn, origin, u, v, lens = cams.device_data()
    rect = worlds.device_data()
    cs = oracle.cam_static(origin, u, v, float(lens))
    st = oracle.seed_states(n * h * w, 0)
    want = oracle.render(dyn, rect, h, w, spp, st, cs=cs)
    ctx.seed(n * h * w, 0, 0)
    ctx.set_scene(dyn, rect, origin, u, v, float(lens))
    got = ctx.render(n, h, w, spp, to_host=True)
    assert np.array_equal(got, want)
    assert np.array_equal(ctx.get_states(0, n * h * w), st)


@pytest.mark.parametrize("lens", [0.6243510725689605, 0.46456785704581477, 0.07, 0.0625],
                         ids=["inexact-1", "inexact-5", "exact", "power-of-two"])
@pytest.mark.parametrize("h", [64, 60])
def test_lens_radius_forms(ctx, oracle, lens, h):
    """offset = float32(float64(p) * lens_radius) (camera.py:343): the canonical-frame kernels
    use a float32 fma form when the host finds it exact for the radius (rf_abi_ctx.hip lens_split)
    and the literal float64 form otherwise -- e.g. for the first two radii here, for which 1 / 5
    of the 25 M possible disc coordinates round differently (tests/test_hostsim.py)."""
    n, spp = 3, 6
    d = helpers.pack_scene(np.array([5.5, 7.0, 9.5], dtype=np.float32), np.array([5.5, 9.0, 6.0], dtype=np.float32))
    cs = oracle.cam_static(d[2], d[3], d[4], lens)
    st = oracle.seed_states(n * h * h, 0)
    want = oracle.render(d[0], d[1], h, h, spp, st, cs=cs)
    ctx.seed(n * h * h, 0, 0)
    ctx.set_scene(d[0], d[1], d[2], d[3], d[4], lens)
    assert np.array_equal(ctx.render(n, h, h, spp, to_host=True), want)
    assert np.array_equal(ctx.get_states(0, n * h * h), st)


@pytest.mark.parametrize("h,w", [(8, 8), (6, 10)])
def test_more_environments_than_one_launch_holds(ctx, oracle, h, w):
    """A launch's grid holds 65 535 environments: larger batches are rendered and scored in chunks
    (launch_render / launch_focus, rf_abi_render.hip).  65 600 environments -- frames, final RNG states and
    focus values of every one of them, on both sides of the chunk boundary, against the oracle; the
    6 x 10 frame takes the byte-store path (w % 4 != 0) with a chunk base that is not dword-aligned."""
    n, spp = 65_600, 2
    rng = np.random.Generator(np.random.PCG64DXSM(65))
    d = helpers.pack_scene(*helpers.random_scene(rng, n))
    states = oracle.seed_states(n * h * w, 0)
    want = oracle.render(d[0], d[1], h, w, spp, states, n_threads=16)
    ctx.seed(n * h * w, 0, 0)
    ctx.set_scene(*d)
    got = ctx.render(n, h, w, spp, to_host=True)
    assert np.array_equal(got[:65_535], want[:65_535]), "first chunk"
    assert np.array_equal(got[65_535:], want[65_535:]), "second chunk"
    assert np.array_equal(ctx.get_states(), states)
    assert np.allclose(ctx.focus(n, h, w), oracle.focus_values(want, n_threads=16), rtol=1e-12, atol=0)


def test_render_extreme_geometry(ctx, oracle):
    """Targets far outside [5, 10]: tiny/huge rectangles, strong defocus, t-range misses."""
    targets = np.array([1.0, 40.0, 0.0005, 2.0e6, 10.0, 5.0], dtype=np.float32)
    focus = np.array([40.0, 1.0, 10.0, 10.0, 0.01, 1.0e4], dtype=np.float32)
    n, h, w, spp = len(targets), 32, 32, 3
    dyn, rect, origin, u, v, lens = helpers.pack_scene(targets, focus)
    st = oracle.seed_states(n * h * w, 0)
    want = oracle.render(dyn, rect, h, w, spp, st)
    ctx.seed(n * h * w, 0, 0)
    ctx.set_scene(dyn, rect, origin, u, v, lens)
    assert np.array_equal(ctx.render(n, h, w, spp, to_host=True), want)
    assert np.array_equal(ctx.get_states(0, n * h * w), st)


def test_partial_render_reuses_low_state_indices(ctx, oracle):
    """Auto-reset renders k < N envs with states 0..k*h*w (vector_environment.py:144,
    render.py:217): the HIP path must consume exactly those streams."""
    n, k, h, spp = 6, 2, 32, 3
    rng = np.random.default_rng(5)
    t, f = helpers.random_scene(rng, n)
    st = oracle.seed_states(n * h * h, 0)
    ctx.seed(n * h * h, 0, 0)
    d = helpers.pack_scene(t, f)
    oracle.render(d[0], d[1], h, h, spp, st)
    ctx.set_scene(*d)
    ctx.render(n, h, h, spp)
    t2, f2 = helpers.random_scene(rng, k)
    d2 = helpers.pack_scene(t2, f2)
    want = oracle.render(d2[0], d2[1], h, h, spp, st)
    ctx.set_scene(*d2)
    got = ctx.render(k, h, h, spp, to_host=True)
    assert np.array_equal(got, want)
    assert np.array_equal(ctx.get_states(), st)


# --- focus -----------------------------------------------------------------------------------


def _focus_frames(n, h, w):
    rng = np.random.default_rng(h * 7 + w)
    frames = rng.integers(0, 256, size=(n, h, w, 3), dtype=np.uint8)
    # one smooth image so that the median/Laplacian see structure, one constant
    yy, xx = np.mgrid[0:h, 0:w]
    frames[0, :, :, 0] = (xx * 3 + yy * 5) % 256
    if n > 1:
        frames[1] = 77
    return frames


# widths that are multiples of 4 (>= 8) take focus_kernel_roll (with and without halo lanes: 64 % (w / 4)), the others the
# byte-per-thread kernel over column tiles of 512; vision.py:11-39 scores whatever it is handed: wide frames too
FOCUS_SHAPES = [(4, 64, 64), (3, 300, 300), (5, 33, 35), (2, 17, 100), (7, 2, 2), (3, 1, 9), (2, 9, 1), (1, 256, 256),
                (2, 16, 16), (2, 15, 20), (1, 600, 600), (3, 70, 8), (2, 5, 12), (2, 37, 128), (1, 100, 512),
                (2, 67, 36), (3, 1, 8), (2, 2, 24), (1, 64, 1024), (1, 40, 2048), (1, 24, 4100), (1, 17, 938),
                (2, 3, 4), (1, 130, 1030)]


@pytest.mark.parametrize("n,h,w", FOCUS_SHAPES)
@pytest.mark.parametrize("gray_mode", [15, 14])
def test_focus_matches_oracle(ctx, oracle, n, h, w, gray_mode):
    frames = _focus_frames(n, h, w)
    ctx.upload_frames(frames)
    got = ctx.focus(n, h, w, gray_mode)
    want = oracle.focus_values(frames, gray_mode)
    assert np.allclose(got, want, rtol=1e-12, atol=1e-12)
    if n > 1:
        assert got[1] == 0.0


@pytest.mark.parametrize("switch", [("REINFOCUS_FOCUS_BAND", "8"), ("REINFOCUS_FOCUS_BAND", "16"),
                                    ("REINFOCUS_FOCUS_BAND", "64"), ("REINFOCUS_FOCUS_BAND", "5"),
                                    ("REINFOCUS_FOCUS_KERNEL", "byte"), ("REINFOCUS_FOCUS_KERNEL", "quad")])
def test_focus_kernels_agree(native, oracle, monkeypatch, switch):
    """Every band height of focus_kernel_roll (the library picks one by launch size), the byte-per-thread kernel and the
    round-2 kernel give the same exact sums: variances equal to the last bit, and equal to the oracle's to 1e-12."""
    base = native.Context(0)
    monkeypatch.setenv(*switch)
    other = native.Context(0)
    try:
        for n, h, w in FOCUS_SHAPES:
            if w % 4 or w < 8:
                continue
            frames = _focus_frames(n, h, w)
            want = oracle.focus_values(frames)
            got = []
            for c in (base, other):
                c.upload_frames(frames)
                got.append(c.focus(n, h, w, 15))
            assert np.array_equal(got[0], got[1]), (n, h, w)
            assert np.allclose(got[0], want, rtol=1e-12, atol=1e-12), (n, h, w)
    finally:
        base.close()
        other.close()


def test_focus_of_a_wide_render(native, oracle):
    """FastRenderer.render(1024) scored on the device (vision.py:28-39 over render.py:127-188): the frame sizes the
    round-5 focus kernels could not take (rows staged whole in 64 KB of LDS)."""
    from reinfocus_amd import vision
    from reinfocus_amd.graphics import render

    r = render.FastRenderer(samples_per_pixel=1, device=0)
    r.update_targets([8.0])
    r.update_focus_planes([8.0])
    frames = r.render(1024)
    got = np.array(vision.focus_values(frames))
    want = oracle.focus_values(np.asarray(frames))
    assert np.allclose(got, want, rtol=1e-12, atol=1e-12)
    r.close()


def test_focus_of_rendered_frames(ctx, oracle):
    n, h, spp = 6, 128, 4
    targets = np.full(n, 8.0, dtype=np.float32)
    focus = np.array([5.0, 6.5, 8.0, 9.0, 10.0, 7.5], dtype=np.float32)
    d = helpers.pack_scene(targets, focus)
    st = oracle.seed_states(n * h * h, 0)
    frames = oracle.render(d[0], d[1], h, h, spp, st, n_threads=8)
    want = oracle.focus_values(frames)
    ctx.seed(n * h * h, 0, 0)
    ctx.set_scene(*d)
    got = ctx.step(n, h, h, spp)
    assert np.max(np.abs(got - want)) < 1e-4          # BASELINE.json tolerance
    assert np.allclose(got, want, rtol=1e-9, atol=1e-9)  # what we actually achieve
    assert np.argmax(got) == 2                         # in focus where focus == target


# --- the Python surface ---------------------------------------------------------------------


def test_fast_renderer_and_vision_surface(native, oracle, kernel_choice):
    """tests/vision_test.py:40-56 of the reference, on the HIP path, plus oracle parity."""
    from reinfocus_amd import vision
    from reinfocus_amd.graphics import render

    r = render.FastRenderer(samples_per_pixel=8, device=0)
    r.update_targets([10] * 5)
    r.update_focus_planes([40, 20, 10, 5, 1])
    frames = r.render(96)
    fv = vision.focus_values(frames)            # device-resident path
    assert fv[2] > fv[3] > fv[4] and fv[2] > fv[1] > fv[0]
    host = np.asarray(frames)
    assert host.shape == (5, 96, 96, 3) and host.dtype == np.uint8
    fv_host = vision.focus_values(host)         # upload path
    assert np.allclose(fv, fv_host, rtol=0, atol=0)
    want = oracle.focus_values(host)
    assert np.allclose(fv, want, rtol=1e-9, atol=1e-9)
    assert abs(vision.focus_value(host[2]) - want[2]) < 1e-9 * max(1.0, want[2])


def test_vision_known_answers(native):
    """tests/vision_test.py:14-34 of the reference."""
    from reinfocus_amd import vision

    assert vision.focus_value(np.zeros((10, 10, 3), dtype=np.uint8)) == 0
    assert vision.focus_value(np.ones((10, 10, 3), dtype=np.uint8)) == 0
    frame = np.zeros((10, 10, 3), dtype=np.uint8)
    frame[0:10:2, :, :] = 255
    frame[:, 0:10:2, :] = 255 - frame[:, 0:10:2, :]
    assert vision.focus_value(frame) > 1


def test_render_known_answers(native, kernel_choice):
    """tests/graphics/render_test.py:83-116 of the reference (FastRendererTest)."""
    from reinfocus_amd.graphics import render

    testee = render.FastRenderer(r_size=30, device=0)
    testee.update_targets([10])
    testee.update_focus_planes([10])
    avg = np.average(np.asarray(testee.render(300)), axis=(0, 1, 2))
    assert np.all(avg >= np.multiply([0.25, 0.25, 0], 255))
    assert np.all(avg <= np.multiply([0.5, 0.5, 0], 255))

    grow = render.FastRenderer(samples_per_pixel=4, device=0)
    grow.update_targets([10])
    grow.update_focus_planes([10])
    grow.render(300)
    grow.render(600)
    result = np.asarray(grow.render(300))
    assert result.shape == (1, 300, 300, 3)
    assert grow._ctx.num_states() == 600 * 600


def test_device_frames_survive_next_render(native):
    from reinfocus_amd.graphics import render

    r = render.FastRenderer(samples_per_pixel=2, device=0)
    r.update_targets([7.0, 8.0])
    r.update_focus_planes([7.0, 8.0])
    a = r.render(32)
    b = r.render(32)
    assert not a.is_resident() and b.is_resident()
    assert np.asarray(a).shape == (2, 32, 32, 3)
    assert not np.array_equal(np.asarray(a), np.asarray(b))  # RNG streams advanced


def test_errors_are_loud(native, ctx):
    with pytest.raises(AssertionError):
        ctx.render(1, 8, 8, 1)  # no scene
    d = helpers.pack_scene([7.0], [7.0])
    ctx.set_scene(*d)
    with pytest.raises(AssertionError):
        ctx.render(1, 8, 8, 1)  # no states
    ctx.seed(64, 0, 0)
    with pytest.raises(AssertionError):
        ctx.render(2, 8, 8, 1)  # n mismatch
    with pytest.raises(AssertionError):
        ctx.render(1, 16, 16, 1)  # not enough states
    ctx.render(1, 8, 8, 1)
    with pytest.raises(AssertionError):
        ctx.focus(1, 16, 16)  # shape mismatch with the frame buffer


def test_fast_sqrt_and_reciprocal_are_ieee_exact(native):
    """Every float in [2^-101, 2^101): the kernel's v_sqrt_f32 + residual fix-up and
    v_rcp_f32 + two fma steps equal the IEEE-correct sqrtf / division (tests/gpucheck)."""
    import ctypes
    import subprocess

    lib = ctypes.CDLL(helpers.built("tests/gpucheck", "libgpucheck.so"))
    out = (ctypes.c_ulonglong * 3)()
    assert lib.gc_check_sqrt_rcp(out) == 0
    assert out[2] > 1_600_000_000          # values visited
    assert out[0] == 0, f"{out[0]} sqrt mismatches"
    assert out[1] == 0, f"{out[1]} reciprocal mismatches"


def test_rejection_candidate_from_subnormal_bits_is_exact(native):
    """approx_pm1 (rf_math.h) reads the top 23 bits of a draw as the bits of a SUBNORMAL float and
    scales it with one fma: every one of the 2^32 high words must give k 2^-22 - 1 exactly on the
    device (no flush-to-zero, no rounding) -- the band analysis of the rejection test rests on it."""
    import ctypes

    lib = ctypes.CDLL(helpers.built("tests/gpucheck", "libgpucheck.so"))
    bad = ctypes.c_ulonglong(0)
    assert lib.gc_check_approx_pm1(ctypes.byref(bad)) == 0
    assert bad.value == 0, f"{bad.value} of 2^32 candidates differ"


def _render_in_child(tmp_path, scene, n, h, spp, env_overrides, w=None, want_kernel=False, replace_env=False):
    """Renders `scene` in a child process whose environment selects another kernel / build of
    the library (the selection is read once, at rf_create); returns frames and final states
    (and the name of the kernel that rendered them)."""
    w = h if w is None else w
    import subprocess
    import sys

    here = os.path.dirname(os.path.abspath(__file__))
    np.savez(tmp_path / "scene.npz", dyn=scene[0], rect=scene[1], origin=scene[2], u=scene[3], v=scene[4],
             lens=scene[5])
    out = tmp_path / "child.npz"
    script = (
        "import sys, numpy as np; sys.path.insert(0, %r)\n"
        "from reinfocus_amd import _native\n"
        "d = np.load(%r)\n"
        "c = _native.Context(0); c.seed(%d, 0, 0)\n"
        "c.set_scene(d['dyn'], d['rect'], d['origin'], d['u'], d['v'], float(d['lens']))\n"
        "f = c.render(%d, %d, %d, %d, to_host=True); s = c.get_states()\n"
        "np.savez(%r, frames=f, states=s, kernel=c.render_kernel_name()); c.close()\n"
    ) % (os.path.dirname(here), str(tmp_path / "scene.npz"), n * h * w, n, h, w, spp, str(out))
    subprocess.check_call([sys.executable, "-c", script],
                          env=dict(env_overrides) if replace_env else dict(os.environ, **env_overrides))
    got = np.load(out)
    if want_kernel:
        return got["frames"], got["states"], str(got["kernel"])
    return got["frames"], got["states"]


@pytest.mark.parametrize("h", [96, 100, 128])
def test_packed_list_overflow_finishes_in_place(ctx, oracle, tmp_path, h):
    """render_kernel_coop2 packs the stragglers of a block into a 256-entry list and lets the
    ones that do not fit finish in their own wave; for power-of-two frames (h = 128) the disc
    stragglers of a wave go into 64 slots of its own instead.  With the production capacities
    those overflow paths are practically never taken, so tests/gpucheck builds the same library
    with a 32-entry list (every all-hit block overflows) and 8 slots per wave (every wave
    overflows), and a child process renders with it: frames and RNG states must equal the
    oracle's, i.e. the production build's."""
    import subprocess

    so = helpers.built("tests/gpucheck", "libreinfocus_cap32.so")
    n, spp = 3, 8
    d = helpers.pack_scene(np.array([5.5, 7.0, 9.5], dtype=np.float32), np.array([5.5, 9.0, 6.0], dtype=np.float32))
    frames, final = _render_in_child(tmp_path, d, n, h, spp, {"REINFOCUS_HIP_LIB": so})
    states = oracle.seed_states(n * h * h, 0)
    want = oracle.render(d[0], d[1], h, h, spp, states)
    assert np.array_equal(frames, want)
    assert np.array_equal(final, states)
    # and the production build agrees, of course
    ctx.seed(n * h * h, 0, 0)
    ctx.set_scene(*d)
    assert np.array_equal(ctx.render(n, h, h, spp, to_host=True), want)


@pytest.mark.parametrize("h", [128, 100, 256])
def test_delayed_waves_change_nothing(oracle, tmp_path, h):
    """The cooperative calls of render_kernel_coop2 are ordered against each other by barriers alone (rf_coop2.h,
    SYNCHRONISATION): no wave may depend on another one being "about as fast".  tests/gpucheck builds the library with
    RF_TEST_SKEW: one wave of every block, a different one from call to call, sleeps ~8 000 cycles before it reads the
    counter after B1, before thread 0's resets and before the collect reads -- long enough for the other waves to be a
    whole phase ahead wherever a barrier does not hold them.  Frames and RNG states must still equal the oracle's.
    (The round-3 form of the call -- no barrier after the collect, one counter -- renders wrong pixels under the same
    delays: profiles/r04_ab.txt section 7.)"""
    so = helpers.built("tests/gpucheck", "libreinfocus_skew.so")
    n, spp = 3, 6
    d = helpers.pack_scene(np.array([5.5, 7.0, 9.5], dtype=np.float32), np.array([5.5, 9.0, 6.0], dtype=np.float32))
    states = oracle.seed_states(n * h * h, 0)
    want = oracle.render(d[0], d[1], h, h, spp, states)  # (advances `states` in place)
    frames, final = _render_in_child(tmp_path, d, n, h, spp, {"REINFOCUS_HIP_LIB": so})
    assert np.array_equal(frames, want)
    assert np.array_equal(final, states)


@pytest.mark.parametrize("n,h,spp,form", [(1, 300, 8, "render_kernel<"), (5, 300, 8, "render_kernel<"), (6, 300, 4, "render_kernel_wave<"),
                                          (8, 300, 8, "render_kernel_wave<"), (24, 300, 2, "render_kernel_wave<"),
                                          (25, 300, 2, "render_kernel_coop2"), (7, 256, 9, "render_kernel<"),
                                          (8, 256, 8, "render_kernel_wave<"), (33, 256, 2, "render_kernel_wave<"),
                                          (34, 256, 2, "render_kernel_coop2"), (2, 128, 3, "render_kernel<"),
                                          (30, 128, 16, "render_kernel<"), (31, 128, 4, "render_kernel_wave<"),
                                          (135, 128, 2, "render_kernel_coop2")])
def test_the_library_picks_the_kernel_by_launch_size(oracle, tmp_path, n, h, spp, form):
    """Without REINFOCUS_RENDER_SETS the library picks the kernel by the pixels of the launch (rf_abi_render.hip
    render_form): up to 500 000 -- the reference's own default, one environment of 300 x 300 at 100 samples, is such a
    launch -- the kernel without cooperative tails (render_kernel: one pixel per thread, no barriers: bound by a sample's
    latency); up to 2.2 M -- examples/ppo_tuned.yml:5: 8 environments -- render_kernel_wave (three sets per wave,
    wave-cooperative tails, no barriers either: about one round of resident waves); beyond, three pixels per thread with
    block-cooperative tails.  Same frames, same RNG states, whichever."""
    rng = np.random.default_rng(n * 100 + h)
    d = helpers.pack_scene(*helpers.random_scene(rng, n))
    states = oracle.seed_states(n * h * h, 0)
    want = oracle.render(d[0], d[1], h, h, spp, states, n_threads=16)  # (advances `states` in place)
    automatic = {k: v for k, v in os.environ.items() if k != "REINFOCUS_RENDER_SETS"}
    frames, final, kernel = _render_in_child(tmp_path, d, n, h, spp, automatic, want_kernel=True, replace_env=True)
    assert kernel.startswith(form), kernel
    assert np.array_equal(frames, want) and np.array_equal(final, states)


@pytest.mark.parametrize("sets", ["w1", "w2", "w3"])
@pytest.mark.parametrize("n,h,w,spp", [(3, 37, 132, 3), (2, 300, 300, 2), (5, 64, 64, 7), (2, 21, 179, 2), (1, 16, 600, 2),
                                       (4, 9, 7, 5), (2, 128, 256, 3), (3, 1, 1, 4), (2, 64, 3, 3)])
def test_wave_cooperative_kernel_matches_oracle(oracle, tmp_path, sets, n, h, w, spp):
    """render_kernel_wave<POW2, LENS, K> for K = 1, 2, 3 sets per wave (REINFOCUS_RENDER_SETS=w1 / w2 / w3; the library
    launches K = 3): frames whose pixel count is and is not a multiple of 4 (dword / byte stores), of 64 K, powers of two
    and not, fewer pixels than one wave has lanes."""
    rng = np.random.default_rng(n * 1000 + w)
    d = helpers.pack_scene(*helpers.random_scene(rng, n))
    states = oracle.seed_states(n * h * w, 0)
    want = oracle.render(d[0], d[1], h, w, spp, states, n_threads=8)  # (advances `states` in place)
    frames, final, kernel = _render_in_child(tmp_path, d, n, h, spp, {"REINFOCUS_RENDER_SETS": sets}, w=w, want_kernel=True)
    assert kernel.startswith("render_kernel_wave<") and (", %s, false>" % sets[1]) in kernel, kernel
    assert np.array_equal(frames, want), kernel
    assert np.array_equal(final, states), kernel


@pytest.mark.parametrize("h", [64, 50])
def test_fallback_render_kernel_matches_oracle(oracle, tmp_path, h):
    """The kernel behind REINFOCUS_RENDER_COOP=0 (render_kernel<AXIS, *> at every size) stays bit-identical to the
    oracle: power-of-two and other frame sizes."""
    n, spp = 4, 5
    overrides = {"REINFOCUS_RENDER_COOP": "0"}
    rng = np.random.default_rng(12)
    d = helpers.pack_scene(*helpers.random_scene(rng, n))
    frames, final = _render_in_child(tmp_path, d, n, h, spp, overrides)
    states = oracle.seed_states(n * h * h, 0)
    assert np.array_equal(frames, oracle.render(d[0], d[1], h, h, spp, states))
    assert np.array_equal(final, states)
