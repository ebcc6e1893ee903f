"""The general renderer on the CPU side: every known answer
the reference's tests hold for spheres, rectangles, worlds, cameras, find_colour and
render (tests/graphics/{sphere,rectangle,world,camera,physics,render,shape_factory}_test.py),
asserted against the oracle and the host mirrors; the numpy-1.26 golden frames; and the
kernel's arithmetic (rf_general.h compiled for the host) against the oracle."""

import ctypes
import os

import numpy as np
import pytest

from tests import helpers
from numpy import testing

HERE = os.path.dirname(os.path.abspath(__file__))
SPHERE, RECTANGLE = 0, 1


def _flatten(hit, rec):
    """numba_test_utils.flatten_hit_result order: hit, p, n, t, uv, uf, m."""
    return np.concatenate([[float(hit)], rec[:12]])


# --- sphere_test.py / rectangle_test.py / world_test.py -------------------------------------


def test_shape_constructors():
    from reinfocus_amd.graphics import shape

    testing.assert_allclose(shape.sphere(shape.v3f(1, 2, 3), 4, shape.v2f(5, 6)).parameters, [1, 2, 3, 4, 5, 6])
    testing.assert_allclose(shape.rectangle(shape.v2f(0, 1), shape.v2f(2, 3), 4, shape.v2f(5, 6)).parameters,
                            [0, 1, 2, 3, 4, 5, 6])
    assert shape.sphere(shape.v3f(), 1).shape_type == shape.SPHERE == 0
    assert shape.rectangle(shape.v2f(), shape.v2f(), 0).shape_type == shape.RECTANGLE == 1


def test_sphere_hit_and_uv(oracle):
    """sphere_test.py:30-66 (exact 13-vector) and :69-90."""
    hit, rec = oracle.sphere_hit([0, 0, 0, 1, 4, 8], (10, 0, 0), (-1, 0, 0), 0.0, 100.0)
    testing.assert_allclose(_flatten(hit, rec), (1, 1, 0, 0, 1, 0, 0, 9, 1, 0.5, 4, 8, SPHERE))
    testing.assert_allclose(oracle.sphere_uv((-1, 0, 0)), (0.0, 0.5), atol=1e-7)
    miss, _ = oracle.sphere_hit([0, 0, 0, 1, 4, 8], (10, 0, 0), (0, 1, 0), 0.0, 100.0)
    assert not miss
    inside, rec = oracle.sphere_hit([0, 0, 0, 1, 4, 8], (0, 0, 0), (0, 0, 1), 0.001, 100.0)
    assert inside and rec[6] == 1.0  # the far root when the near one is behind t_min


def test_rectangle_hit(oracle):
    """rectangle_test.py:35-66 (exact 13-vector)."""
    hit, rec = oracle.rectangle_hit([-1, 1, -1, 1, 1, 4, 8], (0, 0, 0), (0, 0, 1), 0.0, 100.0)
    testing.assert_allclose(_flatten(hit, rec), (1, 0, 0, 1, 0, 0, 1, 1, 0.5, 0.5, 4, 8, RECTANGLE))


def test_world_hit(oracle):
    """world_test.py:131-196, plus the closest-hit rule over two shapes."""
    from reinfocus_amd.graphics import shape, world

    w = world.Worlds([shape.sphere(shape.v3f(0, 0, 0), 1, shape.v2f(4, 8))]).device_data()
    hit, rec = oracle.world_hit(w[0][0], w[1][0], (10, 0, 0), (-1, 0, 0), 0.0, 100.0)
    testing.assert_allclose(_flatten(hit, rec), (1, 1, 0, 0, 1, 0, 0, 9, 1, 0.5, 4, 8, SPHERE))
    w = world.Worlds([shape.rectangle(shape.v2f(-1, 1), shape.v2f(-1, 1), 1, shape.v2f(4, 8))]).device_data()
    hit, rec = oracle.world_hit(w[0][0], w[1][0], (0, 0, 0), (0, 0, 1), 0.0, 100.0)
    testing.assert_allclose(_flatten(hit, rec), (1, 0, 0, 1, 0, 0, 1, 1, 0.5, 0.5, 4, 8, RECTANGLE))
    two = world.Worlds([shape.rectangle(shape.v2f(-1, 1), shape.v2f(-1, 1), 5, shape.v2f(4, 8)),
                        shape.sphere(shape.v3f(0, 0, 2), 0.5, shape.v2f(2, 2))]).device_data()
    hit, rec = oracle.world_hit(two[0][0], two[1][0], (0, 0, 0), (0, 0, 1), 0.0, 100.0)
    assert hit and rec[11] == SPHERE and rec[6] == 1.5  # the nearer sphere wins


def test_worlds_packing():
    """world_test.py:27-102."""
    from reinfocus_amd.graphics import shape, world

    testee = world.Worlds(
        [shape.sphere(shape.v3f(1, 2, 3), 4, shape.v2f(5, 6)),
         shape.rectangle(shape.v2f(-1, 1), shape.v2f(-1, 1), 1, shape.v2f(4, 8))],
        [shape.rectangle(shape.v2f(-0.5, 0.5), shape.v2f(-0.5, 0.5), 0.5, shape.v2f(8, 4))])
    params, types, sizes = testee.device_data()
    assert len(testee) == 2
    testing.assert_allclose(sizes, [2, 1])
    testing.assert_allclose(params, [[[1, 2, 3, 4, 5, 6, 0], [-1, 1, -1, 1, 1, 4, 8]],
                                     [[-0.5, 0.5, -0.5, 0.5, 0.5, 8, 4], [0] * 7]])
    testing.assert_allclose(types, [[SPHERE, RECTANGLE], [RECTANGLE, SPHERE]])


# --- camera_test.py -------------------------------------------------------------------------------


def test_make_gpu_camera_and_cameras():
    """camera_test.py:22-56 (Cameras row) and :114-141 (make_gpu_camera elements)."""
    from reinfocus_amd.graphics import camera

    cam = camera.make_gpu_camera(aperture=2, look_at=(0, 0, -1), vfov=90)
    flat = np.concatenate([np.asarray(cam[k], dtype=np.float64) for k in range(6)] + [[cam[6]]])
    testing.assert_allclose(flat, [-10, -10, -10, 20, 0, 0, 0, 20, 0, 0, 0, 0, 1, 0, 0, 0, 1, 0, 1], atol=1e-5)
    row = camera.Cameras(camera.make_gpu_camera()).device_data()
    assert row.shape == (1, 19) and row.dtype == np.float64
    testing.assert_allclose(row[0], [-2.68, -2.68, -10, 5.36, 0, 0, 0, 5.36, 0, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0.05],
                            atol=0.01)
    assert row[0, 18] == 0.05


# --- shape_factory_test.py -------------------------------------------------------------------------


def test_shape_factory_types():
    from reinfocus_amd.graphics import shape_factory as sf

    def types(shapes):
        return [s.shape_type for s in shapes]

    assert types(sf.one_sphere()) == [SPHERE] and types(sf.two_sphere()) == [SPHERE] * 2
    assert types(sf.one_rect()) == [RECTANGLE] and types(sf.two_rect()) == [RECTANGLE] * 2
    assert set(types(sf.mixed())) == {SPHERE, RECTANGLE}
    assert sf.get_absolute_size(sf.ShapeParameters(size=3.0)) == 3.0
    assert abs(sf.get_absolute_size(sf.ShapeParameters(10.0, r_size=20)) - 1.7632698) < 1e-6


# --- physics_test.py FindColourTest -------------------------------------------------------------------


def test_find_colour(oracle):
    """physics_test.py:175-246: a ray at a rectangle picks up red, at that sphere green."""
    from reinfocus_amd.graphics import shape, world

    w = world.Worlds([shape.rectangle(shape.v2f(-1, 1), shape.v2f(-1, 1), 1)]).device_data()
    st = oracle.seed_states(1, 0)[0]
    col = oracle.find_colour(w[0][0], w[1][0], (-(2**-4), -(2**-4), 0), (-(2**-4), -(2**-4), 1), st)
    assert 0 < col[0] <= 1.0
    testing.assert_allclose(col[1:3], [0, 0])
    w = world.Worlds([shape.sphere(shape.v3f(0, 0, 10), 1)]).device_data()
    st = oracle.seed_states(1, 0)[0]
    col = oracle.find_colour(w[0][0], w[1][0], (-(2**-7), -(2**-6), 0), (-(2**-7), -(2**-6), 1), st)
    assert 0 < col[1] <= 1.0
    testing.assert_allclose(col[::2], [0, 0])


# --- golden frames and the kernel arithmetic --------------------------------------------------------


@pytest.mark.parametrize("name", ["general_small", "general_rect"])
def test_oracle_general_golden(oracle, golden_dir, name):
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    n, h, w, spp = len(g["sizes"]), int(g["h"]), int(g["w"]), int(g["spp"])
    st = oracle.seed_states(n * h * w, 0)
    frames = oracle.render_general(g["cameras"], g["params"], g["types"], g["sizes"], h, w, spp, st, n_threads=4)
    assert np.array_equal(frames, g["frames"]) and np.array_equal(st, g["states_after"])


def _random_scene(rng, n):
    from reinfocus_amd.graphics import camera, shape, world

    cams, envs = [], []
    for _ in range(n):
        cams.append(camera.make_gpu_camera(aperture=rng.uniform(0.05, 0.4), focus_distance=rng.uniform(4, 12),
                                           vfov=rng.uniform(25, 50), aspect_ratio=rng.uniform(0.8, 1.6),
                                           look_from=(rng.uniform(-0.3, 0.3), rng.uniform(-0.3, 0.3), 0.0)))
        shapes = []
        for _ in range(rng.integers(1, 4)):
            z = -rng.uniform(4, 12)
            x, y = rng.uniform(-2, 2), rng.uniform(-1.5, 1.5)
            tex = (int(rng.integers(1, 20)), int(rng.integers(1, 20)))
            if rng.random() < 0.5:
                shapes.append(shape.sphere(shape.v3f(x, y, z), rng.uniform(0.3, 1.5), shape.v2f(*tex)))
            else:
                s = rng.uniform(0.3, 2.0)
                shapes.append(shape.rectangle(shape.v2f(x - s, x + s), shape.v2f(y - s, y + s), z, shape.v2f(*tex)))
        envs.append(shapes)
    return camera.Cameras(*cams).device_data(), world.Worlds(*envs).device_data()


def test_general_kernel_arithmetic_equals_oracle(oracle):
    """rf_general.h compiled for the host (tests/hostsim) on random multi-shape scenes."""
    hs = ctypes.CDLL(helpers.built("tests/hostsim", "libhostsim.so"))
    p = ctypes.c_void_p
    hs.hs_render_general.argtypes = [p] + [ctypes.c_int] * 4 + [p, p, p, p, ctypes.c_int, ctypes.c_int, p]
    _general_kernel_case(oracle, hs, np.random.default_rng(17), 6, 14, 20, 4)
    _general_kernel_case(oracle, hs, np.random.default_rng(18), 4, 16, 32, 5)  # powers of two: float32 coordinates
    _general_kernel_case(oracle, hs, np.random.default_rng(19), 3, 8, 24, 3)   # one of each


def _general_kernel_case(oracle, hs, rng, n, h, w, spp):
    cameras, (params, types, sizes) = _random_scene(rng, n)
    if params.shape[2] < 7:
        params = np.ascontiguousarray(np.pad(params, ((0, 0), (0, 0), (0, 7 - params.shape[2]))))
    st0 = oracle.seed_states(n * h * w, 0)
    st = st0.copy()
    want = oracle.render_general(cameras, params, types, sizes, h, w, spp, st, n_threads=4)
    got = np.zeros_like(want)
    s2 = st0.copy()
    cameras, types, sizes = (np.ascontiguousarray(a) for a in (cameras, types, sizes))
    hs.hs_render_general(got.ctypes.data, n, h, w, spp, cameras.ctypes.data, params.ctypes.data, types.ctypes.data,
                         sizes.ctypes.data, params.shape[1], params.shape[2], s2.ctypes.data)
    assert np.array_equal(got, want) and np.array_equal(s2, st)
    assert len(np.unique(want)) > 20  # the scenes really show something


def test_render_average_colours(oracle):
    """render_test.py:27-80: device_render of one_rect(r_size 30) and render() of
    one_sphere(r_size 30) through make_gpu_camera(), 300 x 300, 100 spp."""
    from reinfocus_amd.graphics import camera, shape_factory as sf, world

    cams = camera.Cameras(camera.make_gpu_camera()).device_data()
    p, t, s = world.Worlds(sf.one_rect(sf.ShapeParameters(r_size=30))).device_data()
    st = oracle.seed_states(300 * 300, 0)
    avg = np.average(oracle.render_general(cams, p, t, s, 300, 300, 100, st, n_threads=8), axis=(0, 1, 2))
    assert np.all(avg >= np.multiply([0.25, 0.25, 0], 255)) and np.all(avg <= np.multiply([0.5, 0.5, 0], 255))
    p, t, s = world.Worlds(sf.one_sphere(sf.ShapeParameters(r_size=30))).device_data()
    p = np.ascontiguousarray(np.pad(p, ((0, 0), (0, 0), (0, 1))))
    st = oracle.seed_states(300 * 300, 0)
    avg = np.average(oracle.render_general(cams, p, t, s, 300, 300, 100, st, n_threads=8), axis=(0, 1, 2))
    assert np.all(avg >= np.multiply([0.4, 0.4, 0.1], 255)) and np.all(avg <= np.multiply([0.6, 0.6, 0.2], 255))


def test_checker_shortcut_equals_the_literal_sine_sign():
    """rf_general.h checker_sign_general (parity of floor(f * u) away from the checker's edges, the
    real sin next to them) against the reference's expression sin((f * pi) * u) (physics.py:58-62)
    with glibc's sin, on random coordinates and on coordinates at / next to every edge."""
    hs = ctypes.CDLL(helpers.built("tests/hostsim", "libhostsim.so"))
    ptr = lambda a: a.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
    n = 2_000_000
    rng = np.random.default_rng(3)
    f = rng.integers(1, 41, n).astype(np.float32)
    edge = (rng.integers(0, 81, n).astype(np.float32) / f).astype(np.float32)
    near = np.maximum(edge.view(np.int32) + rng.integers(-3, 4, n).astype(np.int32), 0).view(np.float32)
    u = np.ascontiguousarray(np.where(rng.random(n) < 0.5, rng.uniform(0, 2, n).astype(np.float32), near))
    literal = np.zeros(n, dtype=np.int32)
    own = np.zeros(n, dtype=np.int32)
    hs.hs_probe_checker_literal(ptr(f), ptr(u), ptr(literal), ctypes.c_uint64(n))
    hs.hs_probe_checker(ptr(f), ptr(u), ptr(own), ctypes.c_uint64(n))
    assert np.array_equal(literal, own)
    assert (literal == 0).sum() > 0 and (literal == 1).sum() > n // 4 and (literal == -1).sum() > n // 4


def _normals_on_and_near_checker_edges(rng, n):
    """Unit normals whose sphere texture coordinates sit on / next to checker edges for the
    frequencies that come with them (70 %), or anywhere (30 %)."""
    fu = rng.integers(1, 41, n).astype(np.float32)
    fv = rng.integers(1, 41, n).astype(np.float32)
    offsets = [0, 0, 1e-9, -1e-9, 1e-7, -1e-7, 3e-7, -3e-7, 1e-6, -1e-6, 1e-5, -1e-5, 1e-4, -1e-4]
    anywhere = rng.random(n) < 0.3
    u = np.where(anywhere, rng.uniform(0, 2, n),
                 np.clip(rng.integers(0, 81, n) / fu.astype(np.float64) + rng.choice(offsets, n), 0, 2))
    v = np.where(anywhere, rng.uniform(0, 1, n),
                 np.clip(rng.integers(0, 41, n) / fv.astype(np.float64) + rng.choice(offsets, n), 0, 1))
    theta, phi = u * np.pi - np.pi, v * np.pi  # atan2(-z, x) = theta, acos(-y) = phi
    ring = np.sin(phi)
    normals = np.stack([ring * np.cos(theta), -np.cos(phi), -ring * np.sin(theta)], axis=1).astype(np.float32)
    return np.ascontiguousarray(normals), fu, fv


def test_sphere_checker_fast_path_equals_the_float64_expressions():
    """rf_general.h sphere_red: float32 approximations of sphere.uv decide the checker colour of a
    sphere hit wherever the products frequency * coordinate are safely away from every integer; the
    reference's float64 atan2 / acos / sin decide otherwise.  (i) the approximations are within
    5e-7 of the float64 values (the margin assumes 1e-6); (ii) fast path + fallback and the float64
    expressions alone give the same colour for normals on and next to every checker edge."""
    hs = ctypes.CDLL(helpers.built("tests/hostsim", "libhostsim.so"))
    ptr = lambda a: a.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
    rng = np.random.default_rng(5)
    n = 2_000_000
    v = rng.normal(size=(n, 3))
    normals = np.ascontiguousarray((v / np.linalg.norm(v, axis=1, keepdims=True)).astype(np.float32))
    normals[: n // 50, 2] = 0.0
    normals[n // 50: n // 25, 0] = 0.0
    normals[n // 25: n // 20, 1] = np.float32(1 - 1e-7) * rng.choice([-1, 1], n // 20 - n // 25)
    approx = np.zeros((n, 2), dtype=np.float32)
    exact = np.zeros((n, 2), dtype=np.float32)
    hs.hs_probe_uv_approx(ptr(normals), ptr(approx), ctypes.c_uint64(n))
    hs.hs_probe_uv(ptr(normals), ptr(exact), ctypes.c_uint64(n))
    error = np.abs(approx.astype(np.float64) - exact)
    error[:, 0] = np.minimum(error[:, 0], np.abs(error[:, 0] - 2))  # u = 0 and u = 2 are the same direction
    assert np.isfinite(exact).all() and error.max() < 5e-7

    n = 4_000_000
    normals, fu, fv = _normals_on_and_near_checker_edges(rng, n)
    fast = np.zeros(n, dtype=np.int32)
    reference = np.zeros(n, dtype=np.int32)
    hs.hs_probe_sphere_red(0, ptr(normals), ptr(fu), ptr(fv), ptr(fast), ctypes.c_uint64(n))
    hs.hs_probe_sphere_red(1, ptr(normals), ptr(fu), ptr(fv), ptr(reference), ctypes.c_uint64(n))
    assert np.array_equal(fast, reference)
    assert 0.4 < reference.mean() < 0.6


def test_sphere_miss_shortcut_equals_the_float64_roots(oracle):
    """rf_general.h sphere_hit answers the certain misses -- a ray pointing away from the centre whose far root is
    below t_min: every ray that has just scattered off the sphere -- from float32 (disc < (b + t_min a)^2 with a
    margin) before any float64 is computed.  Against the reference's float64 expressions (sphere.py:40-103) on rays
    of every kind: leaving the surface at every angle down to grazing, starting just inside / outside it, aimed at
    the sphere from afar, with far roots on both sides of t_min and next to it; hit flags and every bit of the hit
    records must agree, and the shortcut must be what answers most of the leaving rays."""
    hs = ctypes.CDLL(helpers.built("tests/hostsim", "libhostsim.so"))
    ptr = lambda a: a.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
    hs.hs_check_sphere_hit.restype = ctypes.c_long
    rng = np.random.default_rng(11)
    n = 3_000_000
    centre = rng.uniform(-5, 5, (n, 3))
    radius = rng.uniform(0.1, 8, n)
    unit = rng.normal(size=(n, 3))
    unit /= np.linalg.norm(unit, axis=1, keepdims=True)
    kind = rng.integers(0, 4, n)
    # where the ray starts: on the surface (float32 rounding puts it a hair inside or outside), a little off it, or far away
    offset = np.select([kind == 0, kind == 1, kind == 2], [1.0, 1.0 + rng.choice([-1, 1], n) * 10.0 ** rng.uniform(-7, -2, n),
                                                            1.0 + 10.0 ** rng.uniform(-4, 0, n)], rng.uniform(1.5, 6, n))
    origin = centre + unit * (radius * offset)[:, None]
    # direction: the normal plus a vector in / near the unit ball (a scattered ray), or aimed at / past the sphere
    wobble = rng.normal(size=(n, 3))
    wobble *= (rng.uniform(0, 1.05, n) ** (1 / 3) / np.linalg.norm(wobble, axis=1))[:, None]
    aimed = (centre + rng.normal(size=(n, 3)) * radius[:, None] * rng.uniform(0, 1.5, n)[:, None]) - origin
    direction = np.where((kind == 3)[:, None], aimed, unit + wobble)
    direction[: n // 100] *= 1e-4  # tiny and huge directions: the roots scale, the t_min test does not
    direction[n // 100: n // 50] *= 1e4
    spheres = np.ascontiguousarray(np.column_stack([centre, radius]).astype(np.float32))
    origin = np.ascontiguousarray(origin.astype(np.float32))
    direction = np.ascontiguousarray(direction.astype(np.float32))
    hits, shortcuts = ctypes.c_long(0), ctypes.c_long(0)
    for t_min, t_max in ((0.001, 1000000.0), (0.001, 3.0), (0.0, 100.0)):
        bad = hs.hs_check_sphere_hit(ptr(spheres), ptr(origin), ptr(direction), ctypes.c_float(t_min), ctypes.c_float(t_max),
                                     ctypes.c_long(n), ctypes.byref(hits), ctypes.byref(shortcuts))
        assert bad == 0, (t_min, t_max, bad)
        assert n // 10 < hits.value < n and shortcuts.value > n // 4  # both kinds of answer are exercised
    # ... and the literal form the check compares with is the oracle's (a sample through the Python binding)
    for i in range(0, n, n // 400):
        want_hit, want = oracle.sphere_hit(list(spheres[i]) + [4, 8], tuple(origin[i]), tuple(direction[i]), 0.001, 1000000.0)
        from reinfocus_amd.graphics import shape  # noqa: F401  (layout of the record: p, n, t, uv, ...)
        got = hs.hs_check_sphere_hit(ptr(spheres[i:i + 1]), ptr(origin[i:i + 1]), ptr(direction[i:i + 1]), ctypes.c_float(0.001),
                                     ctypes.c_float(1000000.0), ctypes.c_long(1), ctypes.byref(hits), ctypes.byref(shortcuts))
        assert got == 0 and bool(hits.value) == bool(want_hit)


# --- the float32 kernel with abstentions (rf_general_dense.h) ---------------------------------------------------------


def _few_shape_worlds(rng, n, most, kinds="both"):
    """n environments of exactly `most` shapes (spheres and / or rectangles, overlapping in depth and on the screen, some
    spheres large and near the camera) seen by cameras whose axes are the canonical ones -- look_at straight ahead of an
    off-centre look_from -- through the reference's default aperture: the worlds render_general_dense_kernel takes."""
    from reinfocus_amd.graphics import camera, shape, world

    cams, envs = [], []
    for _ in range(n):
        origin = (rng.uniform(-0.5, 0.5), rng.uniform(-0.5, 0.5), rng.uniform(-0.2, 0.2))
        cams.append(camera.make_gpu_camera(focus_distance=rng.uniform(4, 12), vfov=rng.uniform(20, 55),
                                           aspect_ratio=rng.uniform(0.7, 1.8), look_from=origin,
                                           look_at=(origin[0], origin[1], origin[2] - 10.0)))
        shapes = []
        for _ in range(most):
            z = -rng.uniform(3, 12)
            x, y = rng.uniform(-2, 2), rng.uniform(-1.5, 1.5)
            tex = (int(rng.integers(1, 20)), int(rng.integers(1, 20)))
            sphere = rng.random() < 0.5 if kinds == "both" else kinds == "sphere"
            if sphere:
                radius = rng.uniform(0.3, 2.5) if rng.uniform() < 0.9 else rng.uniform(1.0, 1.2) * abs(z)  # (around the camera)
                shapes.append(shape.sphere(shape.v3f(x, y, z), radius, shape.v2f(*tex)))
            else:
                s = rng.uniform(0.3, 2.5)
                shapes.append(shape.rectangle(shape.v2f(x - s, x + s), shape.v2f(y - s, y + s), z, shape.v2f(*tex)))
        envs.append(shapes)
    params, types, sizes = world.Worlds(*envs).device_data()
    if params.shape[2] < 7:
        params = np.pad(params, ((0, 0), (0, 0), (0, 7 - params.shape[2])))
    return (np.ascontiguousarray(camera.Cameras(*cams).device_data(), dtype=np.float64),
            (np.ascontiguousarray(params, dtype=np.float32), np.ascontiguousarray(types, dtype=np.int32),
             np.ascontiguousarray(sizes, dtype=np.int32)))


def _factory_worlds(rng, n, label):
    """The reference's own two-shape factories (shape_factory.py:69-196) at random distances, default cameras."""
    from reinfocus_amd.graphics import camera, shape_factory as sf, world

    make = {"two_sphere": lambda d: sf.two_sphere(sf.ShapeParameters(d + 10), sf.ShapeParameters(d)),
            "two_rect": lambda d: sf.two_rect(sf.ShapeParameters(d + 10), sf.ShapeParameters(d)),
            "mixed": lambda d: sf.mixed(sf.ShapeParameters(d), sf.ShapeParameters(d + 5))}[label]
    params, types, sizes = world.Worlds(*[make(float(d)) for d in rng.uniform(5, 10, n)]).device_data()
    if params.shape[2] < 7:
        params = np.pad(params, ((0, 0), (0, 0), (0, 7 - params.shape[2])))
    cams = camera.Cameras(*[camera.make_gpu_camera(focus_distance=float(f)) for f in rng.uniform(5, 10, n)])
    return (np.ascontiguousarray(cams.device_data(), dtype=np.float64),
            (np.ascontiguousarray(params, dtype=np.float32), np.ascontiguousarray(types, dtype=np.int32),
             np.ascontiguousarray(sizes, dtype=np.int32)))


def test_double_float_sphere_roots_decide_like_the_float64_ones():
    """rf_general_dense.h sphere_hit_dense -- float32 discriminant, the root in double-float, abstention near every
    comparison bound and every float32 rounding boundary -- against the reference's float64 expressions
    (sphere.py:40-103) on rays of every kind (leaving the surface, starting next to it, aimed from afar; tiny and huge
    directions): wherever it does not abstain, the hit flag and every bit of the record are the literal ones'.  The
    device evaluates sqrt and the reciprocals to 1 ulp; here they are correctly rounded and nudged by -1 / 0 / +1 ulp, so
    nothing may depend on more than 1.5 ulp.  The measured error of the double-float root stays below the bound its
    abstention margins are derived from (they are 4x the bound), and rays aimed at a sphere from afar -- the rays of a
    render -- abstain less than once in 10^4."""
    hs = ctypes.CDLL(helpers.built("tests/hostsim", "libhostsim.so"))
    ptr = lambda a: a.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
    hs.hs_check_sphere_hit_dense.restype = ctypes.c_long
    rng = np.random.default_rng(12)
    n = 3_000_000
    centre = rng.uniform(-5, 5, (n, 3))
    radius = rng.uniform(0.1, 8, n)
    unit = rng.normal(size=(n, 3))
    unit /= np.linalg.norm(unit, axis=1, keepdims=True)
    kind = rng.integers(0, 4, n)
    offset = np.select([kind == 0, kind == 1, kind == 2], [1.0, 1.0 + rng.choice([-1, 1], n) * 10.0 ** rng.uniform(-7, -2, n),
                                                            1.0 + 10.0 ** rng.uniform(-4, 0, n)], rng.uniform(1.05, 6, n))
    origin = centre + unit * (radius * offset)[:, None]
    wobble = rng.normal(size=(n, 3))
    wobble *= (rng.uniform(0, 1.0, n) ** (1 / 3) / np.linalg.norm(wobble, axis=1))[:, None]
    aimed = (centre + rng.normal(size=(n, 3)) * radius[:, None] * rng.uniform(0, 1.5, n)[:, None]) - origin
    direction = np.where((kind == 3)[:, None], aimed, unit + wobble)
    direction[: n // 100] *= 1e-4
    direction[n // 100: n // 50] *= 1e4
    spheres = np.ascontiguousarray(np.column_stack([centre, radius]).astype(np.float32))
    origin = np.ascontiguousarray(origin.astype(np.float32))
    direction = np.ascontiguousarray(direction.astype(np.float32))
    counts, err = (ctypes.c_long * 3)(), ctypes.c_double(0)
    for perturb in (0, 4711):
        for t_min, t_max in ((0.001, 1000000.0), (0.001, 3.0)):
            bad = hs.hs_check_sphere_hit_dense(ptr(spheres), ptr(origin), ptr(direction), ctypes.c_float(t_min),
                                               ctypes.c_float(t_max), ctypes.c_long(n), ctypes.c_uint(perturb), counts,
                                               ctypes.byref(err))
            assert bad == 0, (perturb, t_min, t_max, bad)
            assert counts[0] > n // 10 and counts[1] > n // 4, list(counts)  # hits and decided misses are exercised
            assert err.value < 0.8, err.value  # (fraction of the error bound; the margins are 4x the bound)
    aimed_rays = kind == 3
    sp, o, d = (np.ascontiguousarray(a[aimed_rays]) for a in (spheres, origin, direction))
    bad = hs.hs_check_sphere_hit_dense(ptr(sp), ptr(o), ptr(d), ctypes.c_float(0.001), ctypes.c_float(1e6), ctypes.c_long(len(sp)),
                                       ctypes.c_uint(1), counts, ctypes.byref(err))
    assert bad == 0 and counts[2] < len(sp) // 2000, list(counts)


@pytest.mark.parametrize("label,n,h,w,spp", [("two_sphere", 3, 64, 64, 16), ("mixed", 3, 64, 64, 16), ("two_rect", 2, 48, 80, 8),
                                             ("mixed", 2, 50, 70, 60), ("random2", 6, 40, 56, 8), ("random1", 6, 32, 32, 8),
                                             ("random2", 3, 64, 32, 20), ("random3", 5, 48, 48, 8), ("ragged", 8, 40, 56, 8),
                                             ("ragged", 6, 32, 64, 12)])
def test_dense_kernel_arithmetic_equals_oracle_wherever_it_does_not_abstain(oracle, label, n, h, w, spp):
    """rf_general_dense.h render_pixel_dense compiled for the host on the reference's two-shape factories, on random
    worlds of one, two and three shapes under cameras with canonical axes (the float32 lens offset, and the float64 lens
    products on the same scenes) and on ragged worlds under tilted cameras with apertures of every size (_random_scene):
    every pixel that does not abstain has the oracle's bytes and final RNG state (also with the 1-ulp approximations
    nudged), the pixels that abstain are left untouched, and they are few."""
    hs = ctypes.CDLL(helpers.built("tests/hostsim", "libhostsim.so"))
    p = ctypes.c_void_p
    hs.hs_render_general_dense.argtypes = [p] + [ctypes.c_int] * 4 + [p, p, p, p, ctypes.c_int, ctypes.c_int, p, p, ctypes.c_uint,
                                           ctypes.c_int]
    rng = np.random.default_rng(n * 100 + h)
    if label == "ragged":
        cameras, (params, types, sizes) = _random_scene(rng, n)
        cameras, params, types, sizes = (np.ascontiguousarray(a) for a in (cameras, params, types, sizes))
        if params.shape[2] < 7:
            params = np.ascontiguousarray(np.pad(params, ((0, 0), (0, 0), (0, 7 - params.shape[2]))))
    else:
        cameras, (params, types, sizes) = (_few_shape_worlds(rng, n, int(label[-1])) if label.startswith("random")
                                           else _factory_worlds(rng, n, label))
    st0 = oracle.seed_states(n * h * w, 0)
    st = st0.copy()
    want = oracle.render_general(cameras, params, types, sizes, h, w, spp, st, n_threads=8)
    for perturb, simple in ((0, 1), (99, 1), (0, 0)):
        got, s2, gave_up = np.zeros_like(want), st0.copy(), np.zeros(n * h * w, dtype=np.uint8)
        rc = hs.hs_render_general_dense(got.ctypes.data, n, h, w, spp, cameras.ctypes.data, params.ctypes.data, types.ctypes.data,
                                        sizes.ctypes.data, params.shape[1], params.shape[2], s2.ctypes.data, gave_up.ctypes.data,
                                        perturb, simple)
        assert rc == 0
        keep = gave_up.reshape(n, h, w) == 0
        assert np.array_equal(got[keep], want[keep])
        assert np.array_equal(s2.reshape(n, h, w, 2)[keep], st.reshape(n, h, w, 2)[keep])
        assert np.array_equal(s2.reshape(n, h, w, 2)[~keep], st0.reshape(n, h, w, 2)[~keep])
        assert gave_up.mean() < 0.0005 * spp + 0.002, gave_up.mean()  # (about 1e-4 per sample; spheres around the camera more)
    assert len(np.unique(want)) > 20


def test_environments_per_launch_keep_threads_and_pixels_in_32_bits():
    """rf_render_general's listed kernels run on one-dimensional grids of blocks_per_env * ne blocks of 256 threads
    (rf_general_chunk.h): a launch may hold neither more than 2^32 - 1 pixels (the fix-up list's indices) nor more than
    2^32 - 1 THREADS -- HIP rejects such a launch, and padded tiles launch up to 15 % more threads than the frame has
    pixels (round 5 bounded only the pixels: 300 x 300 frames gave chunks of 47 721 environments = 4.41e9 threads)."""
    import ctypes

    lib = ctypes.CDLL(helpers.built("tests/hostsim", "libhostsim.so"))
    lib.hs_general_chunk.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_ulonglong)]
    limit = (1 << 32) - 1
    rng = np.random.default_rng(3)
    shapes = [(300, 300), (256, 256), (16, 16), (1, 1), (7, 9), (4096, 4096), (97, 129), (600, 600), (128, 128), (1, 4000)]
    shapes += [(int(a), int(b)) for a, b in rng.integers(1, 4097, size=(300, 2))]
    for h, w in shapes:
        for kind in (1, 2):
            per_env = ctypes.c_ulonglong(0)
            chunk = lib.hs_general_chunk(kind, h, w, ctypes.byref(per_env))
            assert 1 <= chunk <= 65535
            assert chunk * h * w <= limit or chunk == 1, (h, w, kind)
            assert chunk * per_env.value * 256 <= limit, (h, w, kind)
            assert per_env.value * 256 * (3 if kind == 1 else 1) >= h * w  # every pixel has a thread (one-shape: three per thread)
            # ... and no smaller than it has to be
            assert chunk == 65535 or (chunk + 1) * h * w > limit or (chunk + 1) * per_env.value * 256 > limit, (h, w, kind)
    per_env = ctypes.c_ulonglong(0)
    assert lib.hs_general_chunk(2, 300, 300, ctypes.byref(per_env)) == limit // (361 * 256) and per_env.value == 361
