"""Independent restatement of the reinfocus hot path on numpy-1.26 scalars.

TEST INFRASTRUCTURE (oracle side), not the product.  Run it with an interpreter
whose numpy is 1.26.x (the reference pins numpy~=1.26.4, pyproject.toml:32):

    /opt/conda/bin/python3.9 oracle/gen_golden.py

Purpose: the C oracle (rf_oracle.c) places every float32/float64 rounding point
by hand.  Here nothing is placed by hand: every expression is written with the
operand *types* the reference uses (numpy.float32 scalars, Python floats, Python
ints) and numpy 1.26's legacy scalar promotion -- the same rules numba's typing
follows -- decides where float64 appears.  The two must agree bit for bit; the
outputs are committed under tests/golden/ and checked by tests/test_oracle_golden.py.

The reference cannot be imported in this image (numba, cv2 and gymnasium are
absent), so this file follows its source text; citations are file:line in the
reference checkout.
"""

import math

import numpy as np

assert np.__version__.startswith("1.26"), "needs numpy 1.26 legacy promotion"

f32 = np.float32
MASK = (1 << 64) - 1


# --- numba.cuda.random (third party; numba/cuda/random.py) ------------------------------


def splitmix_state(seed):
    z = (seed + 0x9E3779B97F4A7C15) & MASK
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & MASK
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & MASK
    z = z ^ (z >> 31)
    return [z, z]


def rotl(x, k):
    return ((x << k) | (x >> (64 - k))) & MASK


def nxt(st):
    s0, s1 = st
    out = (s0 + s1) & MASK
    s1 ^= s0
    st[0] = rotl(s0, 55) ^ s1 ^ ((s1 << 14) & MASK)
    st[1] = rotl(s1, 36)
    return out


def jump(st):
    acc0 = acc1 = 0
    for word in (0xBEAC0467EBA5FACB, 0xD86B048B86AA9922):
        for b in range(64):
            if word & (1 << b):
                acc0 ^= st[0]
                acc1 ^= st[1]
            nxt(st)
    st[0], st[1] = acc0, acc1


def seed_states(n, seed):
    states = [splitmix_state(seed)]
    for _ in range(1, n):
        s = list(states[-1])
        jump(s)
        states.append(s)
    return states


def uniform(st):
    # xoroshiro128p_uniform_float32: float32(uint64_to_unit_float64(next))
    return f32(np.float64(nxt(st) >> 11) * (np.float64(1) / np.float64(1 << 53)))


# --- host packing: camera.py:100-179, world.py:100-123, shape_factory.py:29-41 -----------


def pack_cameras(focus_planes, aspect=1, look_from=(0, 0, 0), look_at=(0, 0, -10),
                 up=(0, 1, 0), aperture=0.1, vfov=30):
    lf = tuple(f32(c) for c in look_from)
    la = tuple(f32(c) for c in look_at)
    upv = tuple(f32(c) for c in up)

    def smul(v, s):
        r = np.multiply(v, s)
        return (r[0], r[1], r[2])

    def norm(v):
        return smul(v, 1.0 / float(np.linalg.norm(np.asarray(v))))

    def cross(a, b):
        c = tuple(np.cross(np.asarray(a), np.asarray(b)))
        return (c[0], c[1], c[2])

    half_aperture = np.divide(aperture, 2.0)
    hh = math.tan((vfov * math.pi / 180.0) / 2.0)
    hw = aspect * hh
    d = np.subtract(lf, la)
    w = norm((d[0], d[1], d[2]))
    u = norm(cross(upv, w))
    v = cross(w, u)

    dyn = []
    for fp in np.asarray(focus_planes, dtype=np.float32):
        s = np.sum((smul(u, hw * fp), smul(v, hh * fp), smul(w, fp)), axis=0)
        ll = np.subtract(lf, (s[0], s[1], s[2]))
        dyn.append([(ll[0], ll[1], ll[2]), smul(u, 2.0 * hw * fp), smul(v, 2.0 * hh * fp)])
    return np.array(dyn, dtype=np.float32), lf, u, v, half_aperture


def pack_worlds(targets, r_size=20):
    rows = []
    for target in np.asarray(targets, dtype=np.float32):
        rows.append([target * math.tan(math.radians(r_size / 2)), -target])
    return np.array(rows, dtype=np.float32)


# --- device functions (vector.py, camera.py, rectangle.py, physics.py) -------------------


def v3(x, y, z):
    return (f32(x), f32(y), f32(z))


def add_v3(*vs):
    x = y = z = f32(0.0)
    for v in vs:
        x += v[0]
        y += v[1]
        z += v[2]
    return (x, y, z)


def sub_v3(a, b):
    return (a[0] - b[0], a[1] - b[1], a[2] - b[2])


def smul_v3(v, s):
    return (f32(v[0] * s), f32(v[1] * s), f32(v[2] * s))


def sqlen(v):
    return f32(v[0] ** 2) + f32(v[1] ** 2) + f32(v[2] ** 2)


def unit_disc(st):
    while True:
        a = uniform(st)
        b = uniform(st)
        p = (f32(a) * f32(2.0) - f32(1), f32(b) * f32(2.0) - f32(1))
        if np.less(p[0] * p[0] + p[1] * p[1], 1.0):
            return p


def unit_sphere(st):
    while True:
        a = uniform(st)
        b = uniform(st)
        c = uniform(st)
        p = sub_v3(smul_v3(v3(a, b, c), f32(2.0)), v3(1, 1, 1))
        if np.less(sqlen(p), 1.0):
            return p


def get_ray(cam, s, t, st):
    ll, hor, ver, origin, u, v, lens_radius = cam
    p = unit_disc(st)
    rd = (p[0] * lens_radius, p[1] * lens_radius)
    off = add_v3(origin, smul_v3(u, rd[0]), smul_v3(v, rd[1]))
    return off, sub_v3(add_v3(ll, smul_v3(hor, s), smul_v3(ver, t)), off)


def fast_hit(rect, origin, direction, t_min, t_max):
    radius, z_pos = rect[0], rect[1]
    t = (z_pos - origin[2]) / direction[2]
    if t < t_min or t > t_max:
        return None
    p = add_v3(origin, smul_v3(direction, t))
    if p[0] < -radius or p[0] > radius or p[1] < -radius or p[1] > radius:
        return None
    x_min, x_max, y_min, y_max = -radius, radius, -radius, radius
    uv = (f32((p[0] - x_min) / (x_max - x_min)), f32((p[1] - y_min) / (y_max - y_min)))
    return p, v3(0, 0, 1), f32(t), uv, (f32(32.0), f32(32.0))


def checker(uf, uv):
    si = (uf[0] * math.pi * uv[0], uf[1] * math.pi * uv[1])
    return v3(1, 0, 0) if math.sin(si[0]) * math.sin(si[1]) > 0 else v3(0, 1, 0)


def fast_find_colour(rect, origin, direction, st):
    att = v3(1, 1, 1)
    hit = fast_hit(rect, origin, direction, f32(0.001), f32(1000000.0))
    if hit is not None:
        _, normal, _, uv, uf = hit
        direction = add_v3(normal, unit_sphere(st))
        a = checker(uf, uv)
        att = (att[0] * a[0], att[1] * a[1], att[2] * a[2])
    length = f32(math.sqrt(sqlen(direction)))
    ud = smul_v3(direction, f32(1.0) / length)
    t = 0.5 * (ud[1] + 1.0)
    sky = add_v3(smul_v3(v3(1, 1, 1), 1.0 - t), smul_v3(v3(0.5, 0.7, 1), t))
    return (sky[0] * att[0], sky[1] * att[1], sky[2] * att[2])


def render(cams, rects, h, w, spp, states):
    """render.py:190-246.  Returns (uint8 frames, float32 pre-scale colours)."""
    dyn, origin, u, v, lens_radius = cams
    n = len(rects)
    frames = np.zeros((n, h, w, 3), dtype=np.uint8)
    colours = np.zeros((n, h, w, 3), dtype=np.float32)
    for e in range(n):
        cam = (tuple(dyn[e, 0]), tuple(dyn[e, 1]), tuple(dyn[e, 2]), origin, u, v, lens_radius)
        for y in range(h):
            for x in range(w):
                st = states[e * h * w + y * w + x]
                colour = v3(0, 0, 0)
                for _ in range(spp):
                    s = f32((x + uniform(st)) / w)
                    t = f32((y + uniform(st)) / h)
                    ro, rdir = get_ray(cam, s, t, st)
                    colour = add_v3(colour, fast_find_colour(rects[e], ro, rdir, st))
                colours[e, y, x] = colour
                frames[e, y, x] = smul_v3(colour, f32(255.0 / spp))
    return frames, colours


# ======================================================================================
# General renderer: Cameras / Worlds / device_render /
# find_colour with spheres and rectangles.  render.py:31-119, camera.py:59-91,182-226,
# 255-281, world.py:27-82,126-167, physics.py:95-145, rectangle.py:26-99, sphere.py.
# ======================================================================================

SPHERE, RECTANGLE = 0, 1


def make_gpu_camera(aperture=0.1, aspect_ratio=1, focus_distance=10, look_at=(0, 0, -10),
                    look_from=(0, 0, 0), up=(0, 1, 0), vfov=30):
    lf = tuple(f32(c) for c in look_from)
    la = tuple(f32(c) for c in look_at)
    upv = tuple(f32(c) for c in up)

    def smul(v, s):
        r = np.multiply(v, s)
        return (r[0], r[1], r[2])

    def norm(v):
        return smul(v, 1.0 / float(np.linalg.norm(np.asarray(v))))

    def cross(a, b):
        c = tuple(np.cross(np.asarray(a), np.asarray(b)))
        return (c[0], c[1], c[2])

    hh = math.tan((vfov * math.pi / 180.0) / 2.0)
    hw = aspect_ratio * hh
    d = np.subtract(lf, la)
    w = norm((d[0], d[1], d[2]))
    u = norm(cross(upv, w))
    v = cross(w, u)
    s = np.sum((smul(u, hw * focus_distance), smul(v, hh * focus_distance), smul(w, focus_distance)), axis=0)
    ll = np.subtract(lf, (s[0], s[1], s[2]))
    return ((ll[0], ll[1], ll[2]), smul(u, 2.0 * hw * focus_distance), smul(v, 2.0 * hh * focus_distance),
            lf, u, v, np.divide(aperture, 2.0))


def pack_general_cameras(cams):
    """camera.py:63-83 Cameras: numpy.hstack of the f32 vectors and the f64 lens radius
    (the result is float64[n, 19])."""
    return np.hstack([[c[0] for c in cams], [c[1] for c in cams], [c[2] for c in cams], [c[3] for c in cams],
                      [c[4] for c in cams], [c[5] for c in cams], np.reshape([c[6] for c in cams], (len(cams), 1))])


def sphere_shape(centre, radius, texture=(16, 16)):
    return np.array([*centre, radius, *texture], dtype=np.float32), SPHERE


def rectangle_shape(x_span, y_span, z_pos, texture=(16, 16)):
    return np.array([*x_span, *y_span, z_pos, *texture], dtype=np.float32), RECTANGLE


def pack_worlds_general(env_shapes):
    """world.py:30-65 Worlds."""
    sizes = np.array([len(s) for s in env_shapes], dtype=np.int32)
    most = max(sizes)
    width = max(max(len(p) for p, _ in shapes) for shapes in env_shapes)
    params = np.zeros((len(env_shapes), most, width), dtype=np.float32)
    types = np.zeros((len(env_shapes), most), dtype=np.int32)
    for e, shapes in enumerate(env_shapes):
        for i, (p, t) in enumerate(shapes):
            params[e, i, : len(p)] = p
            types[e, i] = t
    return params, types, sizes


def dot3(a, b):
    return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]


def _acos(x):
    # math.acos raises outside [-1, 1]; on the device (libdevice) the result is NaN
    x = float(x)
    return math.acos(x) if -1.0 <= x <= 1.0 else float("nan")


def sphere_uv(n):
    return (f32((math.atan2(-n[2], n[0]) + math.pi) / math.pi), f32(_acos(-n[1]) / math.pi))


def sphere_hit(p, origin, direction, t_min, t_max):
    centre = v3(p[0], p[1], p[2])
    radius = p[3]
    oc = sub_v3(origin, centre)
    a = dot3(direction, direction)
    b = dot3(oc, direction)
    c = dot3(oc, oc) - radius * radius
    disc = b * b - a * c
    if disc < 0:
        return None
    sqrtd = math.sqrt(disc)
    root = (-b - sqrtd) / a
    if root < t_min or t_max < root:
        root = (-b + sqrtd) / a
        if root < t_min or t_max < root:
            return None
    pt = add_v3(origin, smul_v3(direction, root))
    n = smul_v3(sub_v3(pt, centre), f32(1.0 / radius))
    return pt, n, f32(root), sphere_uv(n), (f32(p[4]), f32(p[5])), f32(SPHERE)


def rectangle_hit(p, origin, direction, t_min, t_max):
    t = (p[4] - origin[2]) / direction[2]
    if t < t_min or t > t_max:
        return None
    pt = add_v3(origin, smul_v3(direction, t))
    x_min, x_max, y_min, y_max = p[0], p[1], p[2], p[3]
    if pt[0] < x_min or pt[0] > x_max or pt[1] < y_min or pt[1] > y_max:
        return None
    uv = (f32((pt[0] - x_min) / (x_max - x_min)), f32((pt[1] - y_min) / (y_max - y_min)))
    return pt, v3(0, 0, 1), f32(t), uv, (f32(p[5]), f32(p[6])), f32(RECTANGLE)


def world_hit(params, types, origin, direction, t_min, t_max):
    closest = t_max
    record = None
    for p, t in zip(params, types):
        h = sphere_hit(p, origin, direction, t_min, closest) if t == SPHERE else \
            rectangle_hit(p, origin, direction, t_min, closest)
        if h is not None:
            closest = h[2]
            record = h
    return record


def checker_general(uf, uv):
    si = (uf[0] * math.pi * uv[0], uf[1] * math.pi * uv[1])
    return v3(1, 0, 0) if math.sin(si[0]) * math.sin(si[1]) > 0 else v3(0, 1, 0)


def find_colour(params, types, origin, direction, st):
    att = v3(1, 1, 1)
    for _ in range(50):
        rec = world_hit(params, types, origin, direction, f32(0.001), f32(1000000.0))
        if rec is not None:
            pt, normal, _, uv, uf, _ = rec
            origin = pt
            direction = add_v3(normal, unit_sphere(st))
            a = checker_general(uf, uv)
            att = (att[0] * a[0], att[1] * a[1], att[2] * a[2])
        else:
            length = f32(math.sqrt(sqlen(direction)))
            ud = smul_v3(direction, f32(1.0) / length)
            t = 0.5 * (ud[1] + 1.0)
            sky = add_v3(smul_v3(v3(1, 1, 1), 1.0 - t), smul_v3(v3(0.5, 0.7, 1), t))
            return (sky[0] * att[0], sky[1] * att[1], sky[2] * att[2])
    return v3(0, 0, 0)


def render_general(cameras, world, h, w, spp, states):
    """render.py:31-85 device_render.  cameras: float64[n, 19]; world: (params, types, sizes)."""
    params, types, sizes = world
    n = len(sizes)
    frames = np.zeros((n, h, w, 3), dtype=np.uint8)
    for e in range(n):
        row = cameras[e]
        cam = (v3(*row[0:3]), v3(*row[3:6]), v3(*row[6:9]), v3(*row[9:12]), v3(*row[12:15]), v3(*row[15:18]),
               row[18])
        k = sizes[e]
        for y in range(h):
            for x in range(w):
                st = states[e * h * w + y * w + x]
                colour = v3(0, 0, 0)
                for _ in range(spp):
                    s = f32((x + uniform(st)) / w)
                    t = f32((y + uniform(st)) / h)
                    ro, rdir = get_ray(cam, s, t, st)
                    colour = add_v3(colour, find_colour(params[e, :k], types[e, :k], ro, rdir, st))
                frames[e, y, x] = smul_v3(colour, f32(255.0 / spp))
    return frames
