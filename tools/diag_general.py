"""Development tool: where does the general renderer's device arithmetic differ from the host's?

CPU part (always): rf_general.h's checker_sign_general against the reference's literal
sin(fl64(fl64(f * pi) * u)) sign on random and adversarial texture coordinates.
GPU part (when a device is visible): the float64 library calls of sphere_hit / sphere.uv and
the checker helper evaluated on the device (tests/gpucheck) and on the host (tests/hostsim, glibc)
for the same operands, bit by bit; then random scenes through rf_render_general against the
oracle, listing the differing pixels.
"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
P = ctypes.c_void_p


def ptr(a):
    return a.ctypes.data_as(P)


def operands(rng, n):
    """(f, u) pairs: uniform ones plus coordinates at and next to the checker's edges."""
    f = rng.integers(1, 21, n).astype(np.float32)
    u = rng.uniform(0, 2, n).astype(np.float32)
    k = rng.integers(0, 41, n).astype(np.float32)
    edge = (k / f).astype(np.float32)
    edge = np.nextafter(edge, np.float32(3) * rng.choice([-1, 1], n).astype(np.float32)).astype(np.float32) \
        if False else edge
    steps = rng.integers(-3, 4, n)
    bits = edge.view(np.int32) + steps.astype(np.int32)
    near = bits.view(np.float32)
    pick = rng.random(n) < 0.5
    u = np.where(pick, u, near).astype(np.float32)
    u[u < 0] = 0
    return f, np.ascontiguousarray(u)


def main():
    hs = ctypes.CDLL(os.path.join(ROOT, "tests", "hostsim", "libhostsim.so"))
    rng = np.random.default_rng(0)
    n = 4_000_000
    f, u = operands(rng, n)
    lit = np.zeros(n, dtype=np.int32)
    own = np.zeros(n, dtype=np.int32)
    hs.hs_probe_checker_literal(ptr(f), ptr(u), ptr(lit), ctypes.c_uint64(n))
    hs.hs_probe_checker(ptr(f), ptr(u), ptr(own), ctypes.c_uint64(n))
    bad = np.flatnonzero(lit != own)
    print(f"host: checker_sign_general vs literal sin sign: {len(bad)} of {n} differ")
    for i in bad[:10]:
        print("   f", f[i], "u", repr(u[i]), "m", float(f[i]) * float(u[i]), "literal", lit[i], "own", own[i])

    from reinfocus_amd import _native
    if _native.device_count() < 1:
        print("no GPU: device part skipped")
        return
    gc = ctypes.CDLL(os.path.join(ROOT, "tests", "gpucheck", "libgpucheck.so"))
    dev = np.zeros(n, dtype=np.int32)
    assert gc.gc_probe_checker(ptr(f), ptr(u), ptr(dev), ctypes.c_uint64(n)) == 0
    bad = np.flatnonzero(dev != lit)
    print(f"device checker_sign_general vs host literal: {len(bad)} of {n} differ")
    for i in bad[:10]:
        print("   f", f[i], "u", repr(u[i]), "m", float(f[i]) * float(u[i]), "literal", lit[i], "device", dev[i])

    # float64 library calls on operands like sphere_hit's
    v = rng.normal(size=(n, 3))
    normals = np.ascontiguousarray((v / np.linalg.norm(v, axis=1, keepdims=True)).astype(np.float32))
    normals[: n // 50, 2] = 0.0  # on the seam of atan2
    normals[n // 50: n // 25, 0] = 0.0
    normals[n // 25: n // 20, 1] = np.float32(1.0) * rng.choice([-1, 1], n // 20 - n // 25)
    a64 = -normals[:, 2].astype(np.float64)
    b64 = normals[:, 0].astype(np.float64)
    c64 = -normals[:, 1].astype(np.float64)
    pos = np.abs(rng.normal(size=n)) * 10.0 ** rng.uniform(-8, 3, n)
    for op, name, a, b in ((0, "sqrt", pos, pos), (1, "div", a64, pos), (2, "atan2", a64, b64), (3, "acos", c64, c64),
                           (4, "sin", (f.astype(np.float64) * np.pi) * u.astype(np.float64), b64),
                           (5, "(atan2+pi)/pi", a64, b64), (6, "acos/pi", c64, c64)):
        a = np.ascontiguousarray(a)
        b = np.ascontiguousarray(b)
        host = np.zeros(n)
        devo = np.zeros(n)
        hs.hs_probe_f64(op, ptr(a), ptr(b), ptr(host), ctypes.c_uint64(n))
        assert gc.gc_probe_f64(op, ptr(a), ptr(b), ptr(devo), ctypes.c_uint64(n)) == 0
        same = (host.view(np.int64) == devo.view(np.int64)) | (np.isnan(host) & np.isnan(devo))
        diff = np.flatnonzero(~same)
        ulps = np.abs(host.view(np.int64)[diff] - devo.view(np.int64)[diff]) if len(diff) else np.array([0])
        f32diff = np.flatnonzero(~((host.astype(np.float32) == devo.astype(np.float32)) | (np.isnan(host) & np.isnan(devo))))
        signdiff = np.flatnonzero(np.sign(host) != np.sign(devo)) if name == "sin" else []
        print(f"f64 {name}: {len(diff)} of {n} differ in the last bits (max {ulps.max()} ulp), {len(f32diff)} after "
              f"the float32 cast, {len(signdiff)} in sign")
        for i in f32diff[:5]:
            print("   a", repr(a[i]), "b", repr(b[i]), "host", repr(host[i]), "device", repr(devo[i]))
    uv_h = np.zeros((n, 2), dtype=np.float32)
    uv_d = np.zeros((n, 2), dtype=np.float32)
    hs.hs_probe_uv(ptr(normals), ptr(uv_h), ctypes.c_uint64(n))
    assert gc.gc_probe_uv(ptr(normals), ptr(uv_d), ctypes.c_uint64(n)) == 0
    same = (uv_h.view(np.int32) == uv_d.view(np.int32)) | (np.isnan(uv_h) & np.isnan(uv_d))
    bad = np.flatnonzero(~same.all(axis=1))
    print(f"sphere uv: {len(bad)} of {n} normals give different float32 (u, v)")
    for i in bad[:10]:
        print("   n", [repr(x) for x in normals[i]], "host", [repr(x) for x in uv_h[i]], "device", [repr(x) for x in uv_d[i]])

    # random scenes
    from oracle import oracle as orc
    from tests.test_general_renderer import _random_scene

    orc.build()
    ctx = _native.Context(0)
    total = 0
    for seed in range(1, 13):
        srng = np.random.default_rng(seed)
        nn, h, w, spp = 8, 48, 56, 8
        cameras, (params, types, sizes) = _random_scene(srng, nn)
        st = orc.seed_states(nn * h * w, 0)
        want = orc.render_general(cameras, params, types, sizes, h, w, spp, st, n_threads=16)
        got = ctx.render_general(cameras, params, types, sizes, h, w, spp)
        d = np.argwhere(np.any(got != want, axis=-1))
        states_equal = np.array_equal(ctx.get_states(0, nn * h * w), st)
        total += len(d)
        print(f"scene seed {seed}: {len(d)} differing pixels, final states equal: {states_equal}")
        for e, y, x in d[:6]:
            print("   env", e, "y", y, "x", x, "want", want[e, y, x], "got", got[e, y, x], "shapes", types[e][: sizes[e]],
                  params[e][: sizes[e]].tolist())
    print("total differing pixels:", total)
    ctx.close()


if __name__ == "__main__":
    main()
