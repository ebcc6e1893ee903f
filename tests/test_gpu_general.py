"""GPU parity of the general renderer (rf_render_general) against the oracle and the
numpy-1.26 golden frames, and the reference's render tests on the HIP path."""

import os

import numpy as np
import pytest

from tests import helpers

from tests.test_general_renderer import _factory_worlds, _few_shape_worlds, _random_scene

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from reinfocus_amd import _native

    assert _native.device_count() >= 1
    c = _native.Context(0)
    yield c
    c.close()


@pytest.fixture()
def ctx_choice(kernel_choice):
    """a context with the session's forced kernels, and one with the library's own choice (conftest.kernel_choice)"""
    from reinfocus_amd import _native

    c = _native.Context(0)
    yield c
    c.close()


@pytest.mark.parametrize("name", ["general_small", "general_rect"])
def test_general_golden(ctx_choice, golden_dir, name):
    ctx = ctx_choice
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    n, h, w, spp = len(g["sizes"]), int(g["h"]), int(g["w"]), int(g["spp"])
    frames = ctx.render_general(g["cameras"], g["params"], g["types"], g["sizes"], h, w, spp)
    assert np.array_equal(frames, g["frames"])
    assert np.array_equal(ctx.get_states(0, n * h * w), g["states_after"])


@pytest.mark.parametrize("n,h,w,spp,seed", [(8, 40, 56, 6, 1), (3, 96, 64, 8, 2), (16, 33, 35, 3, 3), (12, 64, 64, 12, 4),
                                           (6, 50, 128, 5, 5), (2, 256, 256, 4, 6)])
def test_general_random_scenes_match_oracle(ctx_choice, kernel_choice, oracle, n, h, w, spp, seed, monkeypatch):
    """Ragged worlds of one to three shapes under tilted cameras with apertures of every size: the dense kernel's instances
    with float64 lens products -- and the literal kernel (REINFOCUS_GENERAL_DENSE=0) on the same scenes."""
    from reinfocus_amd import _native

    ctx = ctx_choice
    rng = np.random.default_rng(seed)
    cameras, (params, types, sizes) = _random_scene(rng, n)
    st = oracle.seed_states(n * h * w, 0)
    want = oracle.render_general(cameras, params, types, sizes, h, w, spp, st, n_threads=16)
    got = ctx.render_general(cameras, params, types, sizes, h, w, spp)
    # uint8 frames and final RNG states: bit-exact.  float64 atan2 / acos / sin come from the device
    # math library and differ from glibc in the last bits, but reach the result only through a
    # float32 cast or a sign (rf_general.h; test_device_math_library_reaches_the_same_float32 below)
    differing = np.any(got != want, axis=-1).sum()
    assert differing == 0, f"{differing} of {n * h * w} pixels differ"
    assert np.array_equal(ctx.get_states(0, n * h * w), st)
    if kernel_choice == "the library's choice":  # (launches of this size: the literal kernel)
        assert ctx.render_kernel_name().startswith("render_general_kernel")
        return
    assert ctx.render_kernel_name().startswith("render_general_dense_kernel") and ctx.render_kernel_name().endswith(", false>")
    monkeypatch.setenv("REINFOCUS_GENERAL_DENSE", "0")
    literal = _native.Context(0)
    try:
        assert np.array_equal(literal.render_general(cameras, params, types, sizes, h, w, spp), want)
        assert np.array_equal(literal.get_states(0, n * h * w), st)
        assert literal.render_kernel_name().startswith("render_general_kernel")
    finally:
        literal.close()


def test_general_renderer_reseeds_for_every_call(ctx, oracle):
    """render.render creates fresh seed-0 RNG states for every call (graphics/render.py:115).  The
    context remembers the seeded array of the last size and copies it instead of seeding again: repeated
    calls -- with other sizes, other seeds and a fast-path render in between -- must keep giving the
    oracle's frames and final states."""
    rng = np.random.default_rng(31)
    n, h, w, spp = 5, 24, 40, 3
    cameras, (params, types, sizes) = _random_scene(rng, n)
    st = oracle.seed_states(n * h * w, 0)
    want = oracle.render_general(cameras, params, types, sizes, h, w, spp, st, n_threads=8)
    for round_ in range(3):
        got = ctx.render_general(cameras, params, types, sizes, h, w, spp)
        assert np.array_equal(got, want), f"call {round_}"
        assert np.array_equal(ctx.get_states(0, n * h * w), st)
        if round_ == 0:  # another size: the remembered array is replaced ...
            ctx.render_general(cameras[:2], params[:2], types[:2], sizes[:2], 16, 16, 1)
        else:            # ... and here the states are re-seeded differently and advanced by the fast path
            ctx.seed(n * h * w, 7, 3)
            ctx.set_scene(*helpers.pack_scene(np.full(n, 7.0, dtype=np.float32), np.full(n, 6.0, dtype=np.float32)))
            ctx.render(n, h, w, 2)


def test_general_renderer_beyond_one_launch(ctx, oracle):
    """rf_render_general in two chunks (65 535 environments per launch) with a frame whose byte size is
    not a multiple of four (5 x 5 x 3 = 75): the second chunk's frames start at an address that is not
    dword-aligned, so its blocks must take the byte-store path although their offsets within the chunk
    look aligned.  Frames and final RNG states of all 65 540 environments against the oracle."""
    n, h, w, spp = 65_540, 5, 5, 1
    cameras8, (params8, types8, sizes8) = _random_scene(np.random.default_rng(8), 8)
    reps = -(-n // 8)
    cameras = np.ascontiguousarray(np.tile(cameras8, (reps, 1))[:n])
    params = np.ascontiguousarray(np.tile(params8, (reps, 1, 1))[:n])
    types = np.ascontiguousarray(np.tile(types8, (reps, 1))[:n])
    sizes = np.ascontiguousarray(np.tile(sizes8, reps)[:n])
    st = oracle.seed_states(n * h * w, 0)
    want = oracle.render_general(cameras, params, types, sizes, h, w, spp, st, n_threads=16)
    got = ctx.render_general(cameras, params, types, sizes, h, w, spp)
    assert np.array_equal(got[:65_535], want[:65_535]), "first chunk"
    assert np.array_equal(got[65_535:], want[65_535:]), "second chunk"
    assert np.array_equal(ctx.get_states(0, n * h * w), st)


def test_one_shape_worlds_beyond_one_launch(ctx, oracle):
    """The same for worlds of one rectangle (render_general_one_kernel + fix-up list per chunk): 65 540 environments of
    8 x 4 pixels (dword rows: the staged store path) and of 5 x 5 (byte stores), second chunk included."""
    for h, w in ((4, 8), (5, 5)):
        n, spp = 65_540, 2
        cameras8, (params8, types8, sizes8) = _random_one_shape_worlds(np.random.default_rng(9), 8)
        reps = -(-n // 8)
        cameras = np.ascontiguousarray(np.tile(cameras8, (reps, 1))[:n])
        params = np.ascontiguousarray(np.tile(params8, (reps, 1, 1))[:n])
        types = np.ascontiguousarray(np.tile(types8, (reps, 1))[:n])
        sizes = np.ascontiguousarray(np.tile(sizes8, reps)[:n])
        st = oracle.seed_states(n * h * w, 0)
        want = oracle.render_general(cameras, params, types, sizes, h, w, spp, st, n_threads=16)
        got = ctx.render_general(cameras, params, types, sizes, h, w, spp)
        assert ctx.render_kernel_name().startswith("render_general_one_kernel")
        assert np.array_equal(got[:65_535], want[:65_535]), "first chunk"
        assert np.array_equal(got[65_535:], want[65_535:]), "second chunk"
        assert np.array_equal(ctx.get_states(0, n * h * w), st)


def test_reference_render_tests():
    """tests/graphics/render_test.py:27-80 through reinfocus_amd.graphics.render.render."""
    from reinfocus_amd.graphics import camera, render, shape_factory as sf, world

    cams = camera.Cameras(camera.make_gpu_camera())
    frames = render.render(world.Worlds(sf.one_rect(sf.ShapeParameters(r_size=30))), cams, frame_shape=(300, 300),
                           device=0)
    assert frames.shape == (1, 300, 300, 3) and frames.dtype == np.uint8
    avg = np.average(frames, axis=(0, 1, 2))
    assert np.all(avg >= np.multiply([0.25, 0.25, 0], 255)) and np.all(avg <= np.multiply([0.5, 0.5, 0], 255))
    frames = render.render(world.Worlds(sf.one_sphere(sf.ShapeParameters(r_size=30))), cams, frame_shape=(300, 300),
                           device=0)
    avg = np.average(frames, axis=(0, 1, 2))
    assert np.all(avg >= np.multiply([0.4, 0.4, 0.1], 255)) and np.all(avg <= np.multiply([0.6, 0.6, 0.2], 255))
    # default frame shape of the reference's notebooks: 300 x 600, two environments
    frames = render.render(world.Worlds(sf.two_sphere(), sf.mixed()),
                           camera.Cameras(camera.make_gpu_camera(aspect_ratio=2), camera.make_gpu_camera(aspect_ratio=2)),
                           samples_per_pixel=8, device=0)
    assert frames.shape == (2, 300, 600, 3)


def test_general_equals_fast_path_for_one_rectangle(ctx_choice, oracle):
    """One rectangle per env through the general kernel vs the oracle's general path; the
    FastRenderer scene is the same geometry with uf = 32 and a single bounce, so only the
    oracle comparison is exact here."""
    from reinfocus_amd.graphics import camera, shape, world

    cams = camera.Cameras(camera.make_gpu_camera(focus_distance=7.0), camera.make_gpu_camera(focus_distance=9.0))
    worlds = world.Worlds([shape.rectangle(shape.v2f(-1.2, 1.2), shape.v2f(-1.2, 1.2), -7.0, shape.v2f(32, 32))],
                          [shape.rectangle(shape.v2f(-1.6, 1.6), shape.v2f(-1.6, 1.6), -8.0, shape.v2f(32, 32))])
    p, t, s = worlds.device_data()
    st = oracle.seed_states(2 * 64 * 64, 0)
    want = oracle.render_general(cams.device_data(), p, t, s, 64, 64, 5, st, n_threads=8)
    got = ctx_choice.render_general(cams.device_data(), p, t, s, 64, 64, 5)
    assert np.array_equal(got, want)
    assert np.array_equal(ctx_choice.get_states(0, 2 * 64 * 64), st)


def _random_one_shape_worlds(rng, n, kind="rectangle"):
    """n environments of one rectangle (or one sphere) each (off-centre, any size and checker frequencies) seen by cameras
    that look from off the axis through apertures of every size: the worlds render_general_one_kernel takes."""
    from reinfocus_amd.graphics import camera, shape, world

    cams, envs = [], []
    for _ in range(n):
        cams.append(camera.make_gpu_camera(aperture=rng.uniform(0.0, 0.5), focus_distance=rng.uniform(4, 12),
                                           vfov=rng.uniform(20, 55), aspect_ratio=rng.uniform(0.7, 1.8),
                                           look_from=(rng.uniform(-0.5, 0.5), rng.uniform(-0.5, 0.5), rng.uniform(-0.2, 0.2))))
        z = -rng.uniform(3, 12)
        x, y, sx, sy = rng.uniform(-1.5, 1.5), rng.uniform(-1.0, 1.0), rng.uniform(0.3, 3.0), rng.uniform(0.3, 3.0)
        tex = (int(rng.integers(1, 40)), int(rng.integers(1, 40)))
        if kind == "sphere":  # (some of them around the camera: rays that start inside)
            radius = rng.uniform(0.3, 3.0) if rng.uniform() < 0.9 else rng.uniform(14, 20)
            envs.append([shape.sphere(shape.v3f(x, y, z), radius, shape.v2f(*tex))])
            continue
        envs.append([shape.rectangle(shape.v2f(x - sx, x + sx), shape.v2f(y - sy, y + sy), z, shape.v2f(*tex))])
    return camera.Cameras(*cams).device_data(), world.Worlds(*envs).device_data()


@pytest.mark.parametrize("axes", ["tilted", "canonical"])
@pytest.mark.parametrize("kind", ["rectangle", "sphere"])
@pytest.mark.parametrize("n,h,w,spp,seed", [(5, 40, 56, 6, 1), (3, 96, 64, 8, 2), (7, 33, 35, 3, 3), (4, 128, 128, 9, 4),
                                           (2, 300, 300, 4, 5), (3, 16, 260, 5, 6), (2, 7, 500, 2, 7), (6, 64, 64, 20, 8)])
def test_one_shape_worlds_take_the_cooperative_kernel_and_match_the_oracle(ctx, oracle, n, h, w, spp, seed, kind, axes):
    """Worlds of one rectangle per environment go through render_general_one_kernel (rf_general_one.h: the fast path's
    organisation with the general renderer's arithmetic) + the fix-up kernel: frames and final RNG states bit-identical to
    the oracle's general path for power-of-two and other frames, widths that are not multiples of four (byte stores),
    partial tiles, frames wider than high; cameras that look from off the axis through apertures of every size (float64
    lens products) and cameras with canonical axes and the default aperture (the float32 lens offset)."""
    rng = np.random.default_rng(seed)
    cameras, (params, types, sizes) = (_random_one_shape_worlds(rng, n, kind) if axes == "tilted"
                                       else _few_shape_worlds(rng, n, 1, kinds=kind))
    st = oracle.seed_states(n * h * w, 0)
    want = oracle.render_general(cameras, params, types, sizes, h, w, spp, st, n_threads=16)
    got = ctx.render_general(cameras, params, types, sizes, h, w, spp)
    assert ctx.render_kernel_name().startswith("render_general_one_kernel")
    assert (", true, " in ctx.render_kernel_name()) == (kind == "sphere")  # <POW2, SPHERE, WX>
    narrow = -(-w // 64) * 64 < -(-w // 128) * 128  # tiles of 64 x 12 where they leave fewer dead columns
    assert ctx.render_kernel_name().endswith(", 2>" if narrow else ", 4>")
    differing = np.any(got != want, axis=-1).sum()
    assert differing == 0, f"{differing} of {n * h * w} pixels differ"
    assert np.array_equal(ctx.get_states(0, n * h * w), st)


@pytest.mark.parametrize("kind,n,h,w,spp,cooperative", [("rectangle", 2, 300, 600, 3, False), ("rectangle", 36, 256, 256, 2, True),
                                                       ("sphere", 36, 256, 256, 2, False), ("sphere", 50, 256, 256, 2, True)])
def test_small_launches_take_the_literal_kernel(oracle, tmp_path, kind, n, h, w, spp, cooperative):
    """Without REINFOCUS_GENERAL_ONE / _DENSE the library takes the cooperative single-shape kernel and the dense kernel
    only for launches that fill the device (more than 2 M pixels; 3 M for a lone sphere): the notebooks' one or two
    environments are bound by the latency of a pixel's samples, where the literal kernel -- no barriers, no second kernel
    for the pixels that abstained -- is two to three times faster.  Same frames and states either way."""
    import subprocess
    import sys

    rng = np.random.default_rng(n + h)
    cameras, (params, types, sizes) = _random_one_shape_worlds(rng, n, kind)
    np.savez(tmp_path / "scene.npz", cameras=cameras, params=params, types=types, sizes=sizes)
    out = tmp_path / "out.npz"
    script = (
        "import sys; sys.path.insert(0, %r)\n"
        "import numpy as np\n"
        "from reinfocus_amd import _native\n"
        "d = np.load(%r)\n"
        "c = _native.Context(0)\n"
        "f = c.render_general(d['cameras'], d['params'], d['types'], d['sizes'], %d, %d, %d)\n"
        "np.savez(%r, frames=f, states=c.get_states(0, %d), kernel=c.render_kernel_name())\n"
        "c.close()\n"
    ) % (helpers.ROOT, str(tmp_path / "scene.npz"), h, w, spp, str(out), n * h * w)
    automatic = {k: v for k, v in os.environ.items() if k not in ("REINFOCUS_GENERAL_ONE", "REINFOCUS_GENERAL_DENSE")}
    subprocess.check_call([sys.executable, "-c", script], env=automatic)
    got = np.load(out)
    # (36 spheres of 256^2: 2.4 M pixels -- not enough for the cooperative kernel, enough for the dense one)
    other = "render_general_dense_kernel" if n * h * w > 2_000_000 else "render_general_kernel"
    assert str(got["kernel"]).startswith("render_general_one_kernel" if cooperative else other), got["kernel"]
    st = oracle.seed_states(n * h * w, 0)
    want = oracle.render_general(cameras, params, types, sizes, h, w, spp, st, n_threads=16)
    assert np.array_equal(got["frames"], want) and np.array_equal(got["states"], st)


def test_one_shape_worlds_of_both_kinds_do_not_take_the_cooperative_kernel(ctx, oracle):
    """The cooperative kernel is compiled for one kind of shape per launch: a batch whose environments hold one
    rectangle here and one sphere there is rendered by the dense kernel (and equals the oracle as any batch does)."""
    rng = np.random.default_rng(31)
    cam_r, (par_r, typ_r, siz_r) = _random_one_shape_worlds(rng, 2, "rectangle")
    cam_s, (par_s, typ_s, siz_s) = _random_one_shape_worlds(rng, 3, "sphere")
    cameras = np.ascontiguousarray(np.concatenate([cam_r, cam_s]))
    params = np.ascontiguousarray(np.concatenate([par_r, np.pad(par_s, ((0, 0), (0, 0), (0, 1)))]))  # (7 / 6 values per row)
    types = np.ascontiguousarray(np.concatenate([typ_r, typ_s]))
    sizes = np.ascontiguousarray(np.concatenate([siz_r, siz_s]))
    n, h, w, spp = 5, 32, 48, 5
    st = oracle.seed_states(n * h * w, 0)
    want = oracle.render_general(cameras, params, types, sizes, h, w, spp, st, n_threads=16)
    got = ctx.render_general(cameras, params, types, sizes, h, w, spp)
    assert not ctx.render_kernel_name().startswith("render_general_one_kernel")
    assert np.array_equal(got, want) and np.array_equal(ctx.get_states(0, n * h * w), st)


@pytest.mark.parametrize("kind", ["rectangle", "sphere"])
def test_every_abstention_of_the_one_shape_kernel_is_repaired(oracle, tmp_path, kind):
    """The single-rectangle kernel abstains on a pixel whose checker colour float32 cannot decide, or whose scattered ray
    grazes the plane it left, and the fix-up kernel renders those pixels literally.  That happens about once in 10^4
    pixels, and a decision inside the margin is almost always right anyway: an abstention that got lost would go
    unnoticed.  tests/gpucheck builds the library with RF_TEST_DOUBT -- a fifth of the checker decisions and a sixteenth
    of the scattered rays abstain, and whatever abstains is computed WRONG on purpose (inverted sign, mirrored direction)
    -- and a child process renders with it: most pixels are listed, and frames and RNG states must still be the
    oracle's."""
    import subprocess
    import sys

    so = helpers.built("tests/gpucheck", "libreinfocus_doubt.so")
    rng = np.random.default_rng(21)
    out = tmp_path / "out.npz"
    for n, h, w, spp in ((4, 64, 64, 6), (3, 40, 52, 5)):
        cameras, (params, types, sizes) = _random_one_shape_worlds(rng, n, kind)
        np.savez(tmp_path / "scene.npz", cameras=cameras, params=params, types=types, sizes=sizes)
        script = (
            "import sys; sys.path.insert(0, %r)\n"
            "import numpy as np\n"
            "from reinfocus_amd import _native\n"
            "d = np.load(%r)\n"
            "c = _native.Context(0)\n"
            "f = c.render_general(d['cameras'], d['params'], d['types'], d['sizes'], %d, %d, %d)\n"
            "np.savez(%r, frames=f, states=c.get_states(0, %d), redo=c.general_redo_pixels(), kernel=c.render_kernel_name())\n"
            "c.close()\n"
        ) % (helpers.ROOT, str(tmp_path / "scene.npz"), h, w, spp, str(out), n * h * w)
        subprocess.check_call([sys.executable, "-c", script], env=dict(os.environ, REINFOCUS_HIP_LIB=so))
        got = np.load(out)
        st = oracle.seed_states(n * h * w, 0)
        want = oracle.render_general(cameras, params, types, sizes, h, w, spp, st, n_threads=16)
        assert str(got["kernel"]).startswith("render_general_one_kernel")
        assert int(got["redo"]) > n * h * w // 20, "the test build should abstain on many pixels"
        assert np.array_equal(got["frames"], want)
        assert np.array_equal(got["states"], st)


# --- worlds of one or two shapes per environment: the float32 kernel with abstentions (rf_general_dense.h) -------------


def _general_in_child(tmp_path, scene, h, w, spp, env):
    """rf_render_general in a child process (another library, or switches the library reads when a context is created)"""
    import subprocess
    import sys

    cameras, (params, types, sizes) = scene
    np.savez(tmp_path / "scene.npz", cameras=cameras, params=params, types=types, sizes=sizes)
    out = tmp_path / "out.npz"
    script = (
        "import sys; sys.path.insert(0, %r)\n"
        "import numpy as np\n"
        "from reinfocus_amd import _native\n"
        "d = np.load(%r)\n"
        "c = _native.Context(0)\n"
        "f = c.render_general(d['cameras'], d['params'], d['types'], d['sizes'], %d, %d, %d)\n"
        "np.savez(%r, frames=f, states=c.get_states(0, %d), redo=c.general_redo_pixels(), kernel=c.render_kernel_name())\n"
        "c.close()\n"
    ) % (helpers.ROOT, str(tmp_path / "scene.npz"), h, w, spp, str(out), len(sizes) * h * w)
    subprocess.check_call([sys.executable, "-c", script], env=dict(os.environ, **env))
    got = np.load(out)
    return got["frames"], got["states"], int(got["redo"]), str(got["kernel"])


@pytest.mark.parametrize("label,n,h,w,spp", [("two_sphere", 4, 64, 64, 16), ("mixed", 4, 128, 128, 8), ("two_rect", 3, 48, 80, 8),
                                             ("mixed", 2, 300, 300, 12), ("random2", 8, 40, 56, 8), ("random1", 6, 32, 32, 8),
                                             ("random2", 5, 33, 35, 6), ("random2", 3, 256, 256, 5), ("random1", 3, 100, 164, 7),
                                             ("two_sphere", 2, 60, 100, 100), ("random2", 4, 64, 8, 5), ("random2", 6, 7, 90, 4),
                                             ("mixed", 3, 12, 256, 6), ("random3", 5, 48, 64, 7), ("random3", 3, 100, 100, 5)])
def test_few_shape_worlds_take_the_dense_kernel_and_match_the_oracle(ctx, oracle, monkeypatch, label, n, h, w, spp):
    """Worlds of one, two or three shapes in every environment, seen by cameras with canonical axes through an aperture whose
    float32 lens offset is exact, are rendered by render_general_dense_kernel (rf_general_dense.h: no float64, the
    sphere's roots in double-float, pixel-level abstention) + the fix-up kernel: frames and final RNG states
    bit-identical to the oracle's -- power-of-two and other frames, widths that are not multiples of four (byte stores),
    partial tiles, overlapping shapes, spheres around the camera, 100 samples; frames narrower or lower than a 16 x 16 tile,
    which keep the row-major pixel order (full 256-pixel runs through the staged store, partial ones by bytes) -- and the
    literal kernel, asked for with REINFOCUS_GENERAL_DENSE=0, gives the same."""
    from reinfocus_amd import _native

    rng = np.random.default_rng(n * 100 + h)
    cameras, (params, types, sizes) = (_few_shape_worlds(rng, n, int(label[-1])) if label.startswith("random")
                                       else _factory_worlds(rng, n, label))
    st = oracle.seed_states(n * h * w, 0)
    want = oracle.render_general(cameras, params, types, sizes, h, w, spp, st, n_threads=16)
    monkeypatch.setenv("REINFOCUS_GENERAL_ONE", "0")  # (one-shape worlds: not the cooperative kernel of rf_general_one.h)
    monkeypatch.setenv("REINFOCUS_GENERAL_DENSE", "1")
    dense = _native.Context(0)
    monkeypatch.setenv("REINFOCUS_GENERAL_DENSE", "0")
    literal = _native.Context(0)
    try:
        for c, kernel in ((dense, "render_general_dense_kernel"), (literal, "render_general_kernel")):
            got = c.render_general(cameras, params, types, sizes, h, w, spp)
            assert c.render_kernel_name().startswith(kernel), c.render_kernel_name()
            if c is dense:  # 16 x 16 tiles unless they pad the frame more than 15 % beyond what 256-pixel runs do
                tiled = -(-w // 16) * -(-h // 16) * 100 <= -(-h * w // 256) * 115
                assert c.render_kernel_name().endswith(", true, true>" if tiled else ", false, true>"), c.render_kernel_name()
            differing = np.any(got != want, axis=-1).sum()
            assert differing == 0, f"{kernel}: {differing} of {n * h * w} pixels differ"
            assert np.array_equal(c.get_states(0, n * h * w), st), kernel
        assert dense.general_redo_pixels() < n * h * w * (0.0005 * spp + 0.002)
    finally:
        dense.close()
        literal.close()


def test_which_worlds_take_which_dense_instance(ctx, oracle):
    """Up to three shapes per environment, any counts: the dense kernel -- its SIMPLE instances (float32 lens offset) when
    every camera has canonical axes and a lens radius whose float32 offset rf_abi_ctx.hip lens_split finds exact, the
    instances with the reference's float64 lens products for tilted cameras, other radii and launches with more than four
    different radii; four shapes: the literal kernel.  The oracle's frames either way."""
    from reinfocus_amd.graphics import camera, shape_factory as sf, world

    two = sf.two_sphere(sf.ShapeParameters(12.0), sf.ShapeParameters(6.0))
    p, t, s = world.Worlds(two, two).device_data()

    def check(cams, params, types, sizes, kernel, suffix=""):
        st = oracle.seed_states(len(sizes) * 24 * 24, 0)
        want = oracle.render_general(cams.device_data(), params, types, sizes, 24, 24, 3, st, n_threads=4)
        assert np.array_equal(ctx.render_general(cams.device_data(), params, types, sizes, 24, 24, 3), want)
        assert np.array_equal(ctx.get_states(0, len(sizes) * 24 * 24), st)
        name = ctx.render_kernel_name()
        assert name.startswith(kernel) and name.endswith(suffix), name

    straight = camera.Cameras(camera.make_gpu_camera(), camera.make_gpu_camera())
    side = camera.Cameras(camera.make_gpu_camera(look_from=(1.0, 0.5, 0.0)), camera.make_gpu_camera())
    check(straight, p, t, s, "render_general_dense_kernel<false, 2", ", true>")
    check(side, p, t, s, "render_general_dense_kernel<false, 2", ", false>")
    check(straight, *world.Worlds(two + sf.one_rect(), two + sf.one_rect()).device_data(), "render_general_dense_kernel<false, 3", ", true>")
    check(straight, *world.Worlds(two, sf.one_sphere()).device_data(), "render_general_dense_kernel<false, 2", ", true>")  # (ragged)
    check(straight, *world.Worlds(two + two, two + two).device_data(), "render_general_kernel<false>")
    # apertures for which lens_split finds disc coordinates whose float32 offset differs from the float64 one (the radii of
    # tests/test_gpu_parity.py::test_lens_radius_forms), and two for which it finds none
    for aperture, simple in ((2 * 0.6243510725689605, False), (2 * 0.46456785704581477, False), (0.14, True), (0.125, True)):
        cams = camera.Cameras(camera.make_gpu_camera(aperture=aperture), camera.make_gpu_camera(aperture=aperture))
        check(cams, p, t, s, "render_general_dense_kernel<false, 2", ", true>" if simple else ", false>")
    # a proof per radius costs the host ~60 ms: a launch with more than four different apertures is not worth them
    many = camera.Cameras(*[camera.make_gpu_camera(aperture=0.1 + 0.01 * k) for k in range(5)])
    check(many, *world.Worlds(two, two, two, two, two).device_data(), "render_general_dense_kernel<false, 2", ", false>")


def test_a_fix_up_list_that_overflows_is_rendered_again_by_the_literal_kernel(oracle, tmp_path):
    """The fix-up list holds a sixteenth of a launch's pixels (at least 65 536); a launch in which more pixels abstain is
    rendered again, whole, by the literal kernel from the call's fresh seed-0 states.  REINFOCUS_GENERAL_REDO_CAP=1
    makes that happen at test size (a two-sphere scene at 16 samples has a few abstentions per thousand pixels), for
    the dense kernel and for the cooperative one-shape kernel; the call's count is the sum over its launches."""
    rng = np.random.default_rng(3)
    scene = _factory_worlds(rng, 4, "two_sphere")
    n, h, w, spp = 4, 64, 64, 16
    st = oracle.seed_states(n * h * w, 0)
    want = oracle.render_general(scene[0], *scene[1], h, w, spp, st, n_threads=16)
    frames, states, redo, kernel = _general_in_child(tmp_path, scene, h, w, spp, {"REINFOCUS_GENERAL_REDO_CAP": "1"})
    assert kernel.startswith("render_general_dense_kernel") and redo >= 2, (kernel, redo)
    assert np.array_equal(frames, want) and np.array_equal(states, st)
    scene = _random_one_shape_worlds(np.random.default_rng(4), 4, "sphere")
    st = oracle.seed_states(n * h * w, 0)
    want = oracle.render_general(scene[0], *scene[1], h, w, spp, st, n_threads=16)
    frames, states, redo, kernel = _general_in_child(tmp_path, scene, h, w, spp, {"REINFOCUS_GENERAL_REDO_CAP": "1"})
    assert kernel.startswith("render_general_one_kernel") and redo >= 2, (kernel, redo)
    assert np.array_equal(frames, want) and np.array_equal(states, st)


@pytest.mark.parametrize("kind", ["rectangle", "sphere"])
@pytest.mark.parametrize("n,h,w,spp", [(3, 128, 128, 6), (3, 100, 100, 6), (2, 64, 256, 5), (2, 90, 300, 4)])
def test_one_shape_kernel_under_delayed_waves(oracle, tmp_path, kind, n, h, w, spp):
    """render_general_one_kernel orders its cooperative calls against each other by barriers alone, like the fast path's
    kernel whose machinery it uses (rf_coop2.h SYNCHRONISATION; tests/test_sync_model.py checks that both spell the same
    protocol).  The RF_TEST_SKEW build delays one wave of every block -- a different one from call to call -- by ~8 000
    cycles before it reads a call's counter, before thread 0's resets and before the collect reads: frames and RNG
    states must still be the oracle's, for both kinds of shape, with in-wave disc tails (power-of-two frames) and with
    the block-wide disc call (others), in both tile shapes."""
    so = helpers.built("tests/gpucheck", "libreinfocus_skew.so")
    scene = _random_one_shape_worlds(np.random.default_rng(n * 10 + h), n, kind)
    st = oracle.seed_states(n * h * w, 0)
    want = oracle.render_general(scene[0], *scene[1], h, w, spp, st, n_threads=16)
    frames, states, _redo, kernel = _general_in_child(tmp_path, scene, h, w, spp, {"REINFOCUS_HIP_LIB": so, "REINFOCUS_GENERAL_ONE": "1"})
    assert kernel.startswith("render_general_one_kernel"), kernel
    assert np.array_equal(frames, want) and np.array_equal(states, st)


@pytest.mark.parametrize("label", ["two_sphere", "mixed", "random2"])
def test_every_abstention_of_the_dense_kernel_is_repaired(oracle, tmp_path, label):
    """The dense kernel abstains where float32 cannot be proven to give the reference's bits -- a root next to a
    comparison bound or a rounding boundary, a checker colour next to an edge -- about once in 10^3 pixels, and what it
    would have computed there is almost always right anyway: an abstention that got lost would go unnoticed.  The
    RF_TEST_DOUBT build (tests/gpucheck) abstains at a fifth of its decisions and computes whatever abstains WRONG on
    purpose (inverted checker sign, displaced hit point): most pixels are listed -- with the list's default capacity,
    and with one that overflows -- and frames and RNG states must still be the oracle's."""
    so = helpers.built("tests/gpucheck", "libreinfocus_doubt.so")
    rng = np.random.default_rng(22)
    for n, h, w, spp in ((4, 64, 64, 6), (3, 40, 52, 5)):
        scene = _few_shape_worlds(rng, n, 2) if label == "random2" else _factory_worlds(rng, n, label)
        st = oracle.seed_states(n * h * w, 0)
        want = oracle.render_general(scene[0], *scene[1], h, w, spp, st, n_threads=16)
        for env in ({}, {"REINFOCUS_GENERAL_REDO_CAP": "100"}):
            frames, states, redo, kernel = _general_in_child(tmp_path, scene, h, w, spp, dict(env, REINFOCUS_HIP_LIB=so))
            assert kernel.startswith("render_general_dense_kernel"), kernel
            assert redo > n * h * w // 20, "the test build should abstain on many pixels"
            assert np.array_equal(frames, want), env
            assert np.array_equal(states, st), env


def test_dense_kernel_beyond_one_launch(ctx, oracle):
    """65 540 environments of two shapes each in two launches (65 535 environments per launch), 5 x 5 pixel frames (75
    bytes: the second launch's frames start at an address that is not dword-aligned) and 4 x 8 (the staged store path),
    each launch with its own fix-up count."""
    for h, w in ((5, 5), (4, 8)):
        n, spp = 65_540, 2
        cameras8, (params8, types8, sizes8) = _few_shape_worlds(np.random.default_rng(10), 8, 2)
        reps = -(-n // 8)
        cameras = np.ascontiguousarray(np.tile(cameras8, (reps, 1))[:n])
        params = np.ascontiguousarray(np.tile(params8, (reps, 1, 1))[:n])
        types = np.ascontiguousarray(np.tile(types8, (reps, 1))[:n])
        sizes = np.ascontiguousarray(np.tile(sizes8, reps)[:n])
        st = oracle.seed_states(n * h * w, 0)
        want = oracle.render_general(cameras, params, types, sizes, h, w, spp, st, n_threads=16)
        got = ctx.render_general(cameras, params, types, sizes, h, w, spp)
        assert ctx.render_kernel_name().startswith("render_general_dense_kernel")
        assert np.array_equal(got[:65_535], want[:65_535]), "first launch"
        assert np.array_equal(got[65_535:], want[65_535:]), "second launch"
        assert np.array_equal(ctx.get_states(0, n * h * w), st)


def test_device_math_library_reaches_the_same_float32():
    """The operations of the general renderer that go through the device math library, on the
    device (tests/gpucheck) and on the host (tests/hostsim, glibc), same operands: sqrt and '/'
    bit-identical in float64; sphere.uv's float32 (u, v) identical although atan2 / acos differ
    in their last float64 bits; the checker sign identical to the reference's literal sin sign,
    including coordinates at and next to the checker's edges."""
    import ctypes
    here = os.path.dirname(os.path.abspath(__file__))
    gc = ctypes.CDLL(helpers.built("tests/gpucheck", "libgpucheck.so"))
    hs = ctypes.CDLL(helpers.built("tests/hostsim", "libhostsim.so"))
    ptr = lambda a: a.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
    n = 1_000_000
    count = ctypes.c_uint64(n)
    rng = np.random.default_rng(7)
    v = rng.normal(size=(n, 3))
    normals = np.ascontiguousarray((v / np.linalg.norm(v, axis=1, keepdims=True)).astype(np.float32))
    normals[: n // 50, 2] = 0.0           # the seam of atan2
    normals[n // 50: n // 25, 0] = 0.0
    normals[n // 25: n // 20, 1] = rng.choice([-1.0, 1.0], n // 20 - n // 25)  # the poles of acos
    uv_host = np.zeros((n, 2), dtype=np.float32)
    uv_dev = np.zeros((n, 2), dtype=np.float32)
    hs.hs_probe_uv(ptr(normals), ptr(uv_host), count)
    assert gc.gc_probe_uv(ptr(normals), ptr(uv_dev), count) == 0
    assert np.array_equal(uv_host.view(np.int32), uv_dev.view(np.int32))
    positive = np.ascontiguousarray(np.abs(rng.normal(size=n)) * 10.0 ** rng.uniform(-8, 3, n))
    other = np.ascontiguousarray(rng.normal(size=n) * 10.0 ** rng.uniform(-3, 3, n))
    for op in (0, 1):  # sqrt(a), a / b
        host, dev = np.zeros(n), np.zeros(n)
        hs.hs_probe_f64(op, ptr(other if op else positive), ptr(positive), ptr(host), count)
        assert gc.gc_probe_f64(op, ptr(other if op else positive), ptr(positive), ptr(dev), count) == 0
        assert np.array_equal(host.view(np.int64), dev.view(np.int64))
    f = rng.integers(1, 21, n).astype(np.float32)
    edge = (rng.integers(0, 41, n).astype(np.float32) / f).astype(np.float32)
    near = np.maximum((edge.view(np.int32) + rng.integers(-3, 4, n).astype(np.int32)), 0).view(np.float32)
    u = np.ascontiguousarray(np.where(rng.random(n) < 0.5, rng.uniform(0, 2, n).astype(np.float32), near))
    literal = np.zeros(n, dtype=np.int32)
    device = np.zeros(n, dtype=np.int32)
    hs.hs_probe_checker_literal(ptr(f), ptr(u), ptr(literal), count)
    assert gc.gc_probe_checker(ptr(f), ptr(u), ptr(device), count) == 0
    assert np.array_equal(literal, device)


def test_sphere_checker_fast_path_on_the_device():
    """The float32 fast path of a sphere hit's checker colour (rf_general.h sphere_red) as the GPU
    executes it -- v_rcp_f32 / v_sqrt_f32 approximations, fallback through the device math library
    -- against the reference's float64 expressions evaluated on the host with glibc, for normals
    on and next to every checker edge."""
    import ctypes
    from tests.test_general_renderer import _normals_on_and_near_checker_edges

    here = os.path.dirname(os.path.abspath(__file__))
    gc = ctypes.CDLL(helpers.built("tests/gpucheck", "libgpucheck.so"))
    hs = ctypes.CDLL(helpers.built("tests/hostsim", "libhostsim.so"))
    ptr = lambda a: a.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
    n = 4_000_000
    normals, fu, fv = _normals_on_and_near_checker_edges(np.random.default_rng(9), n)
    device = np.zeros(n, dtype=np.int32)
    reference = np.zeros(n, dtype=np.int32)
    assert gc.gc_probe_sphere_red(ptr(normals), ptr(fu), ptr(fv), ptr(device), ctypes.c_uint64(n)) == 0
    hs.hs_probe_sphere_red(1, ptr(normals), ptr(fu), ptr(fv), ptr(reference), ctypes.c_uint64(n))
    assert np.array_equal(device, reference)


def test_an_unknown_lens_radius_is_not_proven_at_once(oracle, tmp_path):
    """The float32 lens offset of the dense kernel's SIMPLE instances needs a 60 ms proof per radius on the host
    (rf_abi_ctx.hip lens_exact_if_known) -- for a render of about a millisecond that is 3 % faster with it.  So a radius
    the process does not know takes the instances with the reference's float64 lens products until it has come back
    (REINFOCUS_LENS_PROVE_AFTER calls: 3 here, 64 by default); the reference's own aperture 0.1 is known from the start.
    The oracle's frames every time."""
    import subprocess
    import sys

    from reinfocus_amd.graphics import camera, shape_factory as sf, world

    two = sf.two_sphere(sf.ShapeParameters(12.0), sf.ShapeParameters(6.0))
    params, types, sizes = world.Worlds(two, two).device_data()
    cams = {"known": camera.Cameras(camera.make_gpu_camera(), camera.make_gpu_camera()).device_data(),
            "new": camera.Cameras(camera.make_gpu_camera(aperture=0.14), camera.make_gpu_camera(aperture=0.14)).device_data()}
    np.savez(tmp_path / "scene.npz", params=params, types=types, sizes=sizes, **cams)
    out = tmp_path / "out.npz"
    script = (
        "import sys; sys.path.insert(0, %r)\n"
        "import numpy as np\n"
        "from reinfocus_amd import _native\n"
        "d = np.load(%r)\n"
        "c = _native.Context(0)\n"
        "names, frames = [], []\n"
        "for which in ('known', 'new', 'new', 'new', 'new'):\n"
        "    frames.append(c.render_general(d[which], d['params'], d['types'], d['sizes'], 24, 24, 3))\n"
        "    names.append(c.render_kernel_name())\n"
        "np.savez(%r, frames=np.stack(frames), names=np.array(names))\n"
        "c.close()\n"
    ) % (helpers.ROOT, str(tmp_path / "scene.npz"), str(out))
    subprocess.check_call([sys.executable, "-c", script], env=dict(os.environ, REINFOCUS_LENS_PROVE_AFTER="3"))
    got = np.load(out)
    names = [str(name) for name in got["names"]]
    assert all(name.startswith("render_general_dense_kernel<false, 2") for name in names), names
    assert [name.endswith(", true>") for name in names] == [True, False, False, True, True], names
    for i, which in enumerate(("known", "new", "new", "new", "new")):
        st = oracle.seed_states(2 * 24 * 24, 0)
        want = oracle.render_general(cams[which], params, types, sizes, 24, 24, 3, st, n_threads=4)
        assert np.array_equal(got["frames"][i], want), (i, names[i])
