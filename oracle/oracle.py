"""ctypes access to the CPU oracle (oracle/librf_oracle.so).

TEST INFRASTRUCTURE, not the product: only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this module.  The product package
(reinfocus_amd) never does.
"""

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "librf_oracle.so")

STATE_DTYPE = np.dtype([("s0", np.uint64), ("s1", np.uint64)], align=True)


class CamStatic(ctypes.Structure):
    _fields_ = [
        ("origin", ctypes.c_float * 3),
        ("u", ctypes.c_float * 3),
        ("v", ctypes.c_float * 3),
        ("lens_radius", ctypes.c_double),
    ]


BUILDS = {"o2": "librf_oracle.so", "o3": "librf_oracle_o3.so"}  # oracle/Makefile: the checker (-O2), and the same source
                                                                  # at -O3 -march=x86-64-v3 for bench.py's cpu_baseline


def build(force=False):
    """Compiles the oracle with gcc (no FMA contraction) where the sources are edited.  On the GPU box
    (/dev/kfd exists) nothing is compiled: the libraries that travelled with the checkout must be the
    builds of the sources next to them (their .srchash, written by the Makefile, must still match)."""
    if os.path.exists("/dev/kfd"):
        import hashlib

        for name in BUILDS.values():
            so = os.path.join(_HERE, name)
            stamp = so + ".srchash"
            assert os.path.exists(so) and os.path.exists(stamp), f"{so} (+ .srchash) missing: run __graft_entry__.build()"
            for line in open(stamp).read().splitlines():
                if line.startswith("flags:"):
                    continue
                digest, path = line.split()
                now = hashlib.sha256(open(os.path.join(_HERE, path), "rb").read()).hexdigest()
                assert now == digest, f"{so} is stale: {path} changed since it was built"
        return _SO
    src = os.path.join(_HERE, "rf_oracle.c")
    for name in BUILDS.values():
        so = os.path.join(_HERE, name)
        if (
            force
            or not os.path.exists(so)
            or not os.path.exists(so + ".srchash")
            or os.path.getmtime(so) < os.path.getmtime(src)
            or os.path.getmtime(so) < os.path.getmtime(os.path.join(_HERE, "rf_oracle.h"))
        ):
            subprocess.check_call(["make", "-C", _HERE, "-B", name])
    return _SO


def use_build(name):
    """Switches every function of this module to another build of the same source ("o2": the checker, the default; "o3":
    bench.py's cpu_baseline times both).  Returns the compiler line of that build."""
    global _SO, _lib
    _SO = os.path.join(_HERE, BUILDS[name])
    _lib = None
    build()
    for line in open(_SO + ".srchash").read().splitlines():
        if line.startswith("flags:"):
            return line[len("flags:"):].strip()
    return "gcc -O2 -std=c11 -fPIC -ffp-contract=off -fno-fast-math -fopenmp"


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = ctypes.CDLL(_SO)
        p = ctypes.c_void_p
        L.orc_seed_states.argtypes = [p, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64]
        L.orc_next.argtypes = [p]
        L.orc_next.restype = ctypes.c_uint64
        L.orc_jump.argtypes = [p]
        L.orc_uniform_float.argtypes = [p]
        L.orc_uniform_float.restype = ctypes.c_float
        L.orc_random_in_unit_disc.argtypes = [p, p]
        L.orc_random_in_unit_sphere.argtypes = [p, p]
        L.orc_get_ray.argtypes = [p, p, ctypes.c_float, ctypes.c_float, p, p, p]
        L.orc_uv.argtypes = [p] + [ctypes.c_float] * 4 + [p]
        L.orc_vector_op.argtypes = [ctypes.c_int, p, p, p, ctypes.c_float, p]
        L.orc_fast_hit.argtypes = [p, p, p, ctypes.c_float, ctypes.c_float, p]
        L.orc_fast_hit.restype = ctypes.c_int
        L.orc_colour_checkerboard.argtypes = [p, p, p]
        L.orc_scatter.argtypes = [p, p, p, p, p]
        L.orc_fast_find_colour.argtypes = [p, p, p, p, p]
        L.orc_render.argtypes = [p] + [ctypes.c_int] * 4 + [p, p, p, p, ctypes.c_int]
        L.orc_sphere_hit.argtypes = [p, p, p, ctypes.c_float, ctypes.c_float, p]
        L.orc_sphere_hit.restype = ctypes.c_int
        L.orc_sphere_uv.argtypes = [p, p]
        L.orc_rectangle_hit.argtypes = [p, p, p, ctypes.c_float, ctypes.c_float, p]
        L.orc_rectangle_hit.restype = ctypes.c_int
        L.orc_world_hit.argtypes = [p, p, ctypes.c_int, ctypes.c_int, p, p, ctypes.c_float, ctypes.c_float, p]
        L.orc_world_hit.restype = ctypes.c_int
        L.orc_find_colour.argtypes = [p, p, ctypes.c_int, ctypes.c_int, p, p, p, p]
        L.orc_render_general.argtypes = [p] + [ctypes.c_int] * 4 + [p, p, p, p, ctypes.c_int, ctypes.c_int, p,
                                                                   ctypes.c_int]
        L.orc_gray.argtypes = [p, ctypes.c_int, ctypes.c_int, ctypes.c_int, p]
        L.orc_median3.argtypes = [p, ctypes.c_int, ctypes.c_int, p]
        L.orc_laplacian_u8.argtypes = [p, ctypes.c_int, ctypes.c_int, p]
        L.orc_var_u8.argtypes = [p, ctypes.c_long]
        L.orc_var_u8.restype = ctypes.c_double
        L.orc_focus_value.argtypes = [p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
        L.orc_focus_value.restype = ctypes.c_double
        L.orc_focus_values.argtypes = [p] + [ctypes.c_int] * 4 + [p, ctypes.c_int]
        _lib = L
    return _lib


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def cam_static(origin=(0, 0, 0), u=(1, 0, 0), v=(0, 1, 0), lens_radius=0.05):
    cs = CamStatic()
    for k in range(3):
        cs.origin[k] = origin[k]
        cs.u[k] = u[k]
        cs.v[k] = v[k]
    cs.lens_radius = float(lens_radius)
    return cs


def seed_states(n, seed=0, subsequence_start=0):
    """graphics/random.py:8-18 make_random_states -> uint64[n, 2] (s0, s1)."""
    st = np.zeros((int(n), 2), dtype=np.uint64)
    lib().orc_seed_states(_ptr(st), int(n), int(seed), int(subsequence_start))
    return st


def next_u64(state):
    return int(lib().orc_next(_ptr(state)))


def uniform_float(state):
    """state: uint64[2] (one row of seed_states), advanced in place."""
    return np.float32(lib().orc_uniform_float(_ptr(state)))


def random_in_unit_disc(state):
    out = np.zeros(2, dtype=np.float32)
    lib().orc_random_in_unit_disc(_ptr(state), _ptr(out))
    return out


def random_in_unit_sphere(state):
    out = np.zeros(3, dtype=np.float32)
    lib().orc_random_in_unit_sphere(_ptr(state), _ptr(out))
    return out


def get_ray(dyn, cs, s, t, state):
    dyn = _f32(dyn).reshape(9)
    o = np.zeros(3, dtype=np.float32)
    d = np.zeros(3, dtype=np.float32)
    lib().orc_get_ray(_ptr(dyn), ctypes.byref(cs), float(s), float(t), _ptr(state), _ptr(o), _ptr(d))
    return o, d


VECTOR_OPS = {"add": 0, "sub": 1, "smul": 2, "dot": 3, "squared_length": 4, "length": 5, "norm": 6,
              "point_at_parameter": 7, "dot2": 8}


def vector_op(name, a, b=(0, 0, 0), c=(0, 0, 0), s=0.0):
    """graphics/vector.py device helpers as the oracle's renderers use them; float32[3]."""
    a, b, c = (_f32(np.resize(np.asarray(v, dtype=np.float32), 3)) for v in (a, b, c))
    out = np.zeros(3, dtype=np.float32)
    lib().orc_vector_op(VECTOR_OPS[name], _ptr(a), _ptr(b), _ptr(c), float(s), _ptr(out))
    return out


def uv(point, x_min, x_max, y_min, y_max):
    pt = _f32(point)
    out = np.zeros(2, dtype=np.float32)
    lib().orc_uv(_ptr(pt), x_min, x_max, y_min, y_max, _ptr(out))
    return out


def fast_hit(rect, origin, direction, t_min, t_max):
    rect, origin, direction = _f32(rect), _f32(origin), _f32(direction)
    rec = np.zeros(13, dtype=np.float32)
    hit = lib().orc_fast_hit(_ptr(rect), _ptr(origin), _ptr(direction), t_min, t_max, _ptr(rec))
    return bool(hit), rec


def colour_checkerboard(uf, uv_):
    uf, uv_ = _f32(uf), _f32(uv_)
    out = np.zeros(3, dtype=np.float32)
    lib().orc_colour_checkerboard(_ptr(uf), _ptr(uv_), _ptr(out))
    return out


def scatter(rec, state):
    rec = _f32(rec)
    o = np.zeros(3, dtype=np.float32)
    d = np.zeros(3, dtype=np.float32)
    a = np.zeros(3, dtype=np.float32)
    lib().orc_scatter(_ptr(rec), _ptr(state), _ptr(o), _ptr(d), _ptr(a))
    return o, d, a


def fast_find_colour(rect, origin, direction, state):
    rect, origin, direction = _f32(rect), _f32(origin), _f32(direction)
    out = np.zeros(3, dtype=np.float32)
    lib().orc_fast_find_colour(_ptr(rect), _ptr(origin), _ptr(direction), _ptr(state), _ptr(out))
    return out


def render(cam_dyn, rect, h, w, spp, states, cs=None, n_threads=1):
    """render.py:190-246 over all pixels; states (uint64[>=n*h*w, 2]) advance in place."""
    cam_dyn = _f32(cam_dyn)
    rect = _f32(rect)
    n = rect.shape[0]
    assert cam_dyn.shape == (n, 3, 3) and rect.shape == (n, 2)
    assert states.shape[0] >= n * h * w and states.dtype == np.uint64
    cs = cs or cam_static()
    frames = np.zeros((n, h, w, 3), dtype=np.uint8)
    lib().orc_render(_ptr(frames), n, h, w, spp, _ptr(cam_dyn), _ptr(rect), ctypes.byref(cs),
                     _ptr(states), n_threads)
    return frames


def sphere_hit(sphere, origin, direction, t_min, t_max):
    sphere, origin, direction = _f32(sphere), _f32(origin), _f32(direction)
    rec = np.zeros(13, dtype=np.float32)
    hit = lib().orc_sphere_hit(_ptr(sphere), _ptr(origin), _ptr(direction), t_min, t_max, _ptr(rec))
    return bool(hit), rec


def sphere_uv(point):
    point = _f32(point)
    out = np.zeros(2, dtype=np.float32)
    lib().orc_sphere_uv(_ptr(point), _ptr(out))
    return out


def rectangle_hit(rect, origin, direction, t_min, t_max):
    rect, origin, direction = _f32(rect), _f32(origin), _f32(direction)
    rec = np.zeros(13, dtype=np.float32)
    hit = lib().orc_rectangle_hit(_ptr(rect), _ptr(origin), _ptr(direction), t_min, t_max, _ptr(rec))
    return bool(hit), rec


def world_hit(params, types, origin, direction, t_min, t_max):
    """params float32[n_shapes, width], types int32[n_shapes] of one environment."""
    params, origin, direction = _f32(params), _f32(origin), _f32(direction)
    types = np.ascontiguousarray(types, dtype=np.int32)
    rec = np.zeros(13, dtype=np.float32)
    hit = lib().orc_world_hit(_ptr(params), _ptr(types), params.shape[0], params.shape[1], _ptr(origin),
                              _ptr(direction), t_min, t_max, _ptr(rec))
    return bool(hit), rec


def find_colour(params, types, origin, direction, state):
    params, origin, direction = _f32(params), _f32(origin), _f32(direction)
    types = np.ascontiguousarray(types, dtype=np.int32)
    out = np.zeros(3, dtype=np.float32)
    lib().orc_find_colour(_ptr(params), _ptr(types), params.shape[0], params.shape[1], _ptr(origin),
                          _ptr(direction), _ptr(state), _ptr(out))
    return out


def render_general(cameras, params, types, sizes, h, w, spp, states, n_threads=1):
    """render.py:31-85 device_render; cameras float64[n,19], params float32[n,most,width]."""
    cameras = np.ascontiguousarray(cameras, dtype=np.float64)
    params = _f32(params)
    types = np.ascontiguousarray(types, dtype=np.int32)
    sizes = np.ascontiguousarray(sizes, dtype=np.int32)
    n, most, width = params.shape
    assert cameras.shape == (n, 19) and types.shape == (n, most) and sizes.shape == (n,)
    frames = np.zeros((n, h, w, 3), dtype=np.uint8)
    lib().orc_render_general(_ptr(frames), n, h, w, spp, _ptr(cameras), _ptr(params), _ptr(types), _ptr(sizes),
                             most, width, _ptr(states), n_threads)
    return frames


def gray(rgb, gray_mode=15):
    rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
    h, w = rgb.shape[:2]
    out = np.zeros((h, w), dtype=np.uint8)
    lib().orc_gray(_ptr(rgb), h, w, gray_mode, _ptr(out))
    return out


def median3(img):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    out = np.zeros_like(img)
    lib().orc_median3(_ptr(img), img.shape[0], img.shape[1], _ptr(out))
    return out


def laplacian_u8(img):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    out = np.zeros_like(img)
    lib().orc_laplacian_u8(_ptr(img), img.shape[0], img.shape[1], _ptr(out))
    return out


def var_u8(img):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    return float(lib().orc_var_u8(_ptr(img), img.size))


def focus_value(rgb, gray_mode=15):
    """vision.py:11-25."""
    rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
    return float(lib().orc_focus_value(_ptr(rgb), rgb.shape[0], rgb.shape[1], gray_mode))


def focus_values(frames, gray_mode=15, n_threads=1):
    """vision.py:28-39."""
    frames = np.ascontiguousarray(frames, dtype=np.uint8)
    n, h, w = frames.shape[:3]
    out = np.zeros(n, dtype=np.float64)
    lib().orc_focus_values(_ptr(frames), n, h, w, gray_mode, _ptr(out), n_threads)
    return out
