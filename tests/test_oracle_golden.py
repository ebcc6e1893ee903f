"""The C oracle against the committed numpy-1.26 golden vectors (tests/golden, written by
oracle/gen_golden.py under numpy 1.26.4) and the host packing logic against the same."""

import os

import numpy as np
import pytest

from tests import helpers


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"))


def test_rng_golden(oracle, golden_dir):
    g = _load(golden_dir, "rng")
    st = oracle.seed_states(6, 0)
    assert np.array_equal(st, g["seeded"])
    s = st[0].copy()
    assert np.array_equal(np.array([oracle.next_u64(s) for _ in range(8)], dtype=np.uint64), g["raw"])
    s = st[3].copy()
    assert np.array_equal(np.array([oracle.uniform_float(s) for _ in range(16)], dtype=np.float32), g["uniform"])
    assert np.array_equal(oracle.seed_states(3, 7), g["seeded_seed7"])
    assert g["top_draw"] == np.float32(1.0)  # uniform_float can return exactly 1.0


def test_splitmix_hand_derived(oracle):
    """splitmix64(0) is a published constant: 0xE220A8397B1DCDAF."""
    st = oracle.seed_states(1, 0)
    assert int(st[0, 0]) == 0xE220A8397B1DCDAF and int(st[0, 1]) == 0xE220A8397B1DCDAF


def test_host_packing_golden(golden_dir):
    from reinfocus_amd.graphics import camera, world

    g = _load(golden_dir, "packing")
    cams = camera.FastCameras()
    cams.update(g["focus_planes"])
    dyn, origin, u, v, lens = cams.device_data()
    assert dyn.dtype == np.float32 and np.array_equal(dyn, g["cam_dyn"])
    assert np.array_equal(origin, g["origin"]) and np.array_equal(u, g["u"]) and np.array_equal(v, g["v"])
    assert isinstance(lens, np.float64) and lens == g["lens_radius"] == 0.05
    for r_size, key in ((20, "rect_r20"), (30, "rect_r30")):
        worlds = world.FastWorlds(r_size)
        worlds.update(g["targets"])
        assert np.array_equal(worlds.device_data(), g[key])


def test_device_functions_golden(oracle, golden_dir):
    g = _load(golden_dir, "device_fns")
    st = oracle.seed_states(4, 0)
    disc = np.array([oracle.random_in_unit_disc(st[0]) for _ in range(8)])
    sphere = np.array([oracle.random_in_unit_sphere(st[1]) for _ in range(8)])
    assert np.array_equal(disc, g["disc"]) and np.array_equal(sphere, g["sphere"])
    cs = oracle.cam_static()
    rays = []
    for s, t in zip(g["st_s"], g["st_t"]):
        o, d = oracle.get_ray(g["cam_dyn"][0], cs, s, t, st[2])
        rays.append(np.concatenate([o, d]))
    assert np.array_equal(np.array(rays), g["rays"])
    cols = [oracle.fast_find_colour(g["rect"], r[0:3], r[3:6], st[3]) for r in g["rays"]]
    assert np.array_equal(np.array(cols), g["colours"])
    assert np.array_equal(st, g["states_after"])


@pytest.mark.parametrize("name", ["render_pow2", "render_npot", "render_rsize30", "render_cfg1", "render_mid"])
def test_render_golden(oracle, golden_dir, name):
    g = _load(golden_dir, name)
    h, w, spp = int(g["h"]), int(g["w"]), int(g["spp"])
    dyn, rect, origin, u, v, lens = helpers.pack_scene(g["targets"], g["focus"], float(g["r_size"]))
    st = oracle.seed_states(len(g["targets"]) * h * w, 0)
    for p in range(int(g["passes"])):
        frames = oracle.render(dyn, rect, h, w, spp, st, cs=oracle.cam_static(origin, u, v, lens))
        assert np.array_equal(frames, g["frames%d" % p])
        # the u8 frame is the truncation of the golden float colour
        scale = np.float32(255.0 / spp)
        assert np.array_equal((g["colours%d" % p] * scale).astype(np.uint8), frames)
    assert np.array_equal(st, g["states_after"])


def test_render_threads_do_not_change_results(oracle):
    rng = np.random.default_rng(0)
    t, f = helpers.random_scene(rng, 5)
    d = helpers.pack_scene(t, f)
    a = oracle.seed_states(5 * 24 * 24, 0)
    b = a.copy()
    fa = oracle.render(d[0], d[1], 24, 24, 3, a, n_threads=1)
    fb = oracle.render(d[0], d[1], 24, 24, 3, b, n_threads=8)
    assert np.array_equal(fa, fb) and np.array_equal(a, b)


def test_the_o3_build_of_the_oracle_gives_the_same_bits(oracle, golden_dir):
    """bench.py's cpu_baseline also times the oracle at -O3 -march=x86-64-v3 (oracle/Makefile: librf_oracle_o3.so, parity
    flags intact): the same frames, RNG states and focus values as the checker's -O2 build -- the numpy-1.26 goldens, a
    random scene, and the general renderer's arithmetic."""
    from tests.test_general_renderer import _random_scene

    rng = np.random.default_rng(5)
    t, f = helpers.random_scene(rng, 4)
    d = helpers.pack_scene(t, f)
    cameras, (params, types, sizes) = _random_scene(np.random.default_rng(6), 3)
    results = {}
    try:
        for build in ("o2", "o3"):
            flags = oracle.use_build(build)
            assert "-ffp-contract=off" in flags and "-fno-fast-math" in flags, flags
            for name in ("render_pow2", "render_npot", "render_mid"):
                g = _load(golden_dir, name)
                h, w, spp = int(g["h"]), int(g["w"]), int(g["spp"])
                dyn, rect, origin, u, v, lens = helpers.pack_scene(g["targets"], g["focus"], float(g["r_size"]))
                st = oracle.seed_states(len(g["targets"]) * h * w, 0)
                for p in range(int(g["passes"])):
                    frames = oracle.render(dyn, rect, h, w, spp, st, cs=oracle.cam_static(origin, u, v, lens))
                    assert np.array_equal(frames, g["frames%d" % p]), (build, name)
                assert np.array_equal(st, g["states_after"]), (build, name)
            st = oracle.seed_states(4 * 50 * 70, 0)
            frames = oracle.render(d[0], d[1], 50, 70, 9, st, n_threads=4)
            st2 = oracle.seed_states(3 * 40 * 33, 0)
            general = oracle.render_general(cameras, params, types, sizes, 40, 33, 5, st2, n_threads=4)
            results[build] = (frames, st.copy(), oracle.focus_values(frames), general, st2.copy())
    finally:
        oracle.use_build("o2")
    for a, b in zip(results["o2"], results["o3"]):
        assert np.array_equal(a, b)
