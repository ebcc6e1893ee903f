"""Parity at BASELINE.json's full sizes.  The oracle cannot render 4096 x 256^2 x 16 spp in
seconds, so the full-size runs are checked through size-independent properties of the
domain -- environments are independent given their RNG states -- plus bit-exact spot checks
of randomly chosen environments against the oracle:

 * spot check: any environment of the full render == the oracle's render of that environment
   alone, started from the same 65536 RNG states (and the states end identical);
 * shard invariance: rendering envs [a, b) on a context seeded at first_state_index = a*h*w
   reproduces the corresponding slice of the single-context render (the multi-GPU contract);
 * determinism: same seed, same scene -> same checksum; focus is invariant under image flips
   (gray is pointwise, median / Laplacian kernels are symmetric, var is permutation invariant).
"""

import zlib

import numpy as np
import pytest

from tests import helpers

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def native():
    from reinfocus_amd import _native

    assert _native.device_count() >= 1
    return _native


def _scene(n, seed):
    rng = np.random.Generator(np.random.PCG64DXSM(seed))
    return helpers.pack_scene(*helpers.random_scene(rng, n))


def _spot_check(oracle, ctx, scene, n, h, spp, picks, states_before, frames_getter):
    dyn, rect, origin, u, v, lens = scene
    cs = oracle.cam_static(origin, u, v, lens)
    for e in picks:
        st = states_before[e].copy()
        want = oracle.render(dyn[e:e + 1], rect[e:e + 1], h, h, spp, st, cs=cs, n_threads=16)
        got = frames_getter(e)
        assert np.array_equal(got, want[0]), f"env {e} differs from the oracle"
        assert np.array_equal(ctx.get_states(e * h * h, h * h), st), f"env {e} states differ"


def test_headline_config_spot_checks_and_focus(native, oracle):
    """BASELINE configs[2]: 4096 envs x 256 x 256 x 16 spp on one GPU."""
    n, h, spp = 4096, 256, 16
    scene = _scene(n, 42)
    ctx = native.Context(0)
    ctx.seed(n * h * h, 0, 0)
    picks = [0, 1, 777, 2048, 4095]
    before = {e: ctx.get_states(e * h * h, h * h) for e in picks}
    ctx.set_scene(*scene)
    fv = ctx.step(n, h, h, spp)
    assert fv.shape == (n,) and np.all(np.isfinite(fv)) and np.all(fv >= 0)

    def frame(e):
        out = np.empty((1, h, h, 3), dtype=np.uint8)
        native._check(native.load().rf_get_frames(ctx._h, e, 1, out.ctypes.data_as(native.ctypes.c_void_p)))
        return out[0]

    _spot_check(oracle, ctx, scene, n, h, spp, picks, before, frame)
    want_fv = oracle.focus_values(np.stack([frame(e) for e in picks]))
    assert np.max(np.abs(fv[picks] - want_fv)) < 1e-4           # BASELINE tolerance
    assert np.allclose(fv[picks], want_fv, rtol=1e-12, atol=0)   # achieved

    # in-focus environments score higher than strongly defocused ones (sanity at scale)
    gap = np.abs(scene[1][:, 1] - scene[0][:, 0, 2])  # |(-target) - (-focus)|
    assert fv[gap < 0.25].mean() > 2 * fv[gap > 3.0].mean()
    ctx.close()


def test_shard_invariance_and_determinism(native):
    """Two 'GPUs' of 1024 envs each == one context of 2048 envs (BASELINE configs[3] logic)."""
    n, h, spp = 2048, 128, 4
    scene = _scene(n, 7)
    dyn, rect, origin, u, v, lens = scene
    whole = native.Context(0)
    whole.seed(n * h * h, 0, 0)
    whole.set_scene(*scene)
    frames = whole.render(n, h, h, spp, to_host=True)
    fv = whole.focus(n, h, h)
    checksum = zlib.crc32(frames.tobytes())

    half = n // 2
    for r in range(2):
        shard = native.Context(0)
        shard.seed(half * h * h, 0, r * half * h * h)
        shard.set_scene(dyn[r * half:(r + 1) * half], rect[r * half:(r + 1) * half], origin, u, v, lens)
        part = shard.render(half, h, h, spp, to_host=True)
        assert np.array_equal(part, frames[r * half:(r + 1) * half])
        assert np.array_equal(shard.focus(half, h, h), fv[r * half:(r + 1) * half])
        assert np.array_equal(shard.get_states(), whole.get_states(r * half * h * h, half * h * h))
        shard.close()

    again = native.Context(0)
    again.seed(n * h * h, 0, 0)
    again.set_scene(*scene)
    assert zlib.crc32(again.render(n, h, h, spp, to_host=True).tobytes()) == checksum
    again.close()

    # flips leave every focus value unchanged, exactly
    whole.upload_frames(np.ascontiguousarray(frames[:256, ::-1]))
    assert np.array_equal(whole.focus(256, h, h), fv[:256])
    whole.upload_frames(np.ascontiguousarray(frames[:256, :, ::-1]))
    assert np.array_equal(whole.focus(256, h, h), fv[:256])
    whole.close()


def test_high_fidelity_config_spot_check(native, oracle):
    """BASELINE configs[4] per-GPU share: 128 envs x 512 x 512 x 64 spp (1024 envs over 8 GPUs)."""
    n, h, spp = 128, 512, 64
    scene = _scene(n, 99)
    ctx = native.Context(0)
    ctx.seed(n * h * h, 0, 0)
    picks = [3, 127]
    before = {e: ctx.get_states(e * h * h, h * h) for e in picks}
    ctx.set_scene(*scene)
    fv = ctx.step(n, h, h, spp)

    def frame(e):
        out = np.empty((1, h, h, 3), dtype=np.uint8)
        native._check(native.load().rf_get_frames(ctx._h, e, 1, out.ctypes.data_as(native.ctypes.c_void_p)))
        return out[0]

    _spot_check(oracle, ctx, scene, n, h, spp, picks, before, frame)
    want = oracle.focus_values(np.stack([frame(e) for e in picks]))
    assert np.allclose(fv[picks], want, rtol=1e-12, atol=0)
    ctx.close()


def test_first_hip_config(native, oracle):
    """BASELINE configs[1]: 256 envs x 128 x 128 x 4 spp, every environment against the oracle."""
    n, h, spp = 256, 128, 4
    scene = _scene(n, 1)
    ctx = native.Context(0)
    ctx.seed(n * h * h, 0, 0)
    ctx.set_scene(*scene)
    st = oracle.seed_states(n * h * h, 0)
    want = oracle.render(scene[0], scene[1], h, h, spp, st, n_threads=16)
    got = ctx.render(n, h, h, spp, to_host=True)
    assert np.array_equal(got, want)
    assert np.array_equal(ctx.get_states(), st)
    assert np.allclose(ctx.focus(n, h, h), oracle.focus_values(want, n_threads=16), rtol=1e-12, atol=0)
    ctx.close()


@pytest.mark.parametrize("n,h,spp,branch", [(4096, 256, 16, "fused-graph"), (4096, 256, 16, "count-sized"),
                                            (128, 512, 64, "fused-graph"), (128, 512, 64, "graph")])
def test_benchmarked_environment_equals_host_harness_at_full_size(n, h, spp, branch, monkeypatch):
    """The object bench.py measures, at the sizes it is measured at (BASELINE configs[2] and the
    per-GPU share of configs[4]): harness.DeviceVectorDiscreteSteps -- whose rf_env_step takes the
    fused schedule (one render launch whose blocks of the re-rendered slots make two passes), and with
    REINFOCUS_ENV_FUSED=0 the count-sized branch at the headline size (rf_env_step_begin, one host round trip,
    rf_env_step_end; 128 environments of 512 x 512 are few enough blocks for the one-sync / hipGraph
    schedule) -- against harness.VectorDiscreteSteps, the reference's numpy glue
    (environments/vector_environment.py:104-164) around rf_render / rf_focus, which the tests above
    and tests/test_gpu_parity.py pin to the oracle.  With a time limit of 4 steps the diverging rule
    ends some environments in step 3 (a partial render of a few), the time limit the rest in step 4
    (a partial render of most), and from then on the episodes are out of step with each other;
    observations, rewards, flags and states must be identical bit for bit over 7 steps, and both
    initializers must have consumed the same draws."""
    from reinfocus_amd.environments import harness

    kw = dict(max_episode_steps=4, num_envs=n, frame_height=h, samples_per_pixel=spp, seed=3, device=0)
    host = harness.VectorDiscreteSteps(**kw)
    if not branch.startswith("fused"):
        monkeypatch.setenv("REINFOCUS_ENV_FUSED", "0")
    dev = harness.DeviceVectorDiscreteSteps(**kw)
    monkeypatch.delenv("REINFOCUS_ENV_FUSED", raising=False)
    o_h, _ = host.reset()
    o_d, _ = dev.reset()
    assert np.array_equal(o_h, o_d) and np.array_equal(host._state, dev._state)
    rng = np.random.default_rng(17)
    ended = []
    for step in range(7):
        actions = rng.integers(0, 13, n)
        want, got = host.step(actions), dev.step(actions)
        first = {"graph": "one-sync", "fused-graph": "fused"}.get(branch, branch)
        assert dev._ctx.env_last_step_branch() == (first if step == 0 else branch)
        for a, b in zip(want[:4], got[:4]):
            assert a.dtype == b.dtype and np.array_equal(a, b)
        assert np.array_equal(host._state, dev._state)
        ended.append(int(got[3].sum()))
    # steps 1-2: nobody can have ended; step 3: only the diverging rule; step 4: everybody else
    assert ended[:2] == [0, 0] and 0 < ended[2] < n and ended[2] + ended[3] == n and 0 < ended[6] < n
    assert host._initializer._generator.bit_generator.state == dev._initializer._generator.bit_generator.state
    # the RNG states both renderers hold are the same 16 B x n x h x h (spot-checked at both ends)
    for first in (0, (n - 1) * h * h):
        assert np.array_equal(host._renderer._ctx.get_states(first, h * h), dev._ctx.get_states(first, h * h))
    host.close()
    dev.close()


@pytest.mark.parametrize("fused", ["1", "0"])
def test_environment_step_beyond_one_launch(fused, monkeypatch):
    """More environments than one launch holds (65 535 rows of blocks): the fused step's render is then several
    launches whose two-pass test needs the launch's first environment (RenderArgs::env0), and its focus launch of
    2 n rows several more (FocusArgs::row0); the count of environments that end crosses the launch boundary in
    step 4 (everybody ends: time limit).  Against the numpy glue around rf_render / rf_focus, bit for bit."""
    from reinfocus_amd.environments import harness

    n, h = 65600, 16
    kw = dict(max_episode_steps=4, num_envs=n, frame_height=h, samples_per_pixel=1, seed=8, device=0)
    host = harness.VectorDiscreteSteps(**kw)
    monkeypatch.setenv("REINFOCUS_ENV_FUSED", fused)
    dev = harness.DeviceVectorDiscreteSteps(**kw)
    monkeypatch.delenv("REINFOCUS_ENV_FUSED")
    assert np.array_equal(host.reset()[0], dev.reset()[0])
    rng = np.random.default_rng(23)
    ended = []
    for _step in range(6):
        actions = rng.integers(0, 13, n)
        want, got = host.step(actions), dev.step(actions)
        for a, b in zip(want[:4], got[:4]):
            assert a.dtype == b.dtype and np.array_equal(a, b)
        assert np.array_equal(host._state, dev._state)
        ended.append(int(got[3].sum()))
    assert dev._ctx.env_last_step_branch() == ("fused-graph" if fused == "1" else "count-sized")
    assert 0 < ended[2] < n and ended[2] + ended[3] == n and 0 < ended[5] < n
    for first in (0, 65535 * h * h, (n - 1) * h * h):
        assert np.array_equal(host._renderer._ctx.get_states(first, h * h), dev._ctx.get_states(first, h * h))
    host.close()
    dev.close()


@pytest.mark.parametrize("n,h,spp,shards", [(32768, 256, 16, 8), (1024, 512, 64, 8)])
def test_whole_multi_gpu_configuration_on_eight_contexts(n, h, spp, shards):
    """BASELINE configs[3] (32768 envs x 256 x 256 x 16 spp) and configs[4] (1024 envs x 512 x 512 x 64
    spp) WHOLE, as the product object runs them on an 8-GPU node -- harness.ShardedVectorDiscreteSteps,
    one context + one host thread per shard -- rehearsed with all eight contexts on this one device
    (34 GB of RNG states: the 288 GB of one MI355X hold the node's whole workload).  Every shard's slice
    must equal a single context seeded at that shard's first_state_index (global pixel index = e h w +
    y w + x up to 2^31 - 1, graphics/render.py:217) and fed the rows of the ONE global initializer in
    global index order (environments/vector_environment.py:138-142): reset, steps without ends, a step
    in which the diverging rule ends a few environments of every shard, the step in which the time limit
    ends the rest, and one more."""
    import copy

    from reinfocus_amd.environments import harness

    kw = dict(max_episode_steps=4, frame_height=h, samples_per_pixel=spp)
    many = harness.ShardedVectorDiscreteSteps(num_envs=n, devices=[0] * shards, seed=5, **kw)
    assert many._ranges == [(g * (n // shards), n // shards) for g in range(shards)]
    assert [s.first_state_index for s in many._shards] == [g * (n // shards) * h * h for g in range(shards)]
    twin = harness._Initializer((5.0, 10.0), 5)  # the global initializer, replayed
    obs0, _ = many.reset()
    state0 = many._state
    assert np.array_equal(state0, twin.initialize(n))
    rng = np.random.default_rng(23)
    record = []
    for _ in range(5):
        actions = rng.integers(0, 13, n)
        pool = copy.deepcopy(twin._generator).uniform(5.0, 10.0, size=(n, 2)).astype(np.float32)
        obs, rewards, _, truncated, _ = many.step(actions)
        twin.initialize(int(truncated.sum())) if truncated.any() else None
        record.append((actions, pool, obs, rewards, truncated, many._state))
    ended = [int(r[4].sum()) for r in record]
    assert ended[:2] == [0, 0] and 0 < ended[2] < n // 8 and ended[2] + ended[3] == n
    assert twin._generator.bit_generator.state == many._initializer._generator.bit_generator.state
    final_states = [shard.ctx.get_states(0, h * h) for shard in many._shards]
    many.close()

    for g, (first, count) in enumerate(many._ranges):
        rows = slice(first, first + count)
        single = harness.DeviceVectorDiscreteSteps(num_envs=count, seed=0, device=0, first_state_index=first * h * h, **kw)
        o, _ = single.reset(state=state0[rows])
        assert np.array_equal(o, obs0[rows]), f"shard {g}: reset"
        for step, (actions, pool, obs, rewards, truncated, state) in enumerate(record):
            r, t, k = single._ctx.env_step_begin(actions[rows])
            before = int(truncated[:first].sum())  # environments that ended in the shards before this one
            assert k == int(truncated[rows].sum())
            got = single._ctx.env_step_end(pool[before:before + k])
            assert np.array_equal(r, rewards[rows]) and np.array_equal(t, truncated[rows]), f"shard {g} step {step}"
            assert np.array_equal(got, obs[rows]), f"shard {g} step {step}: observations"
            assert np.array_equal(single._state, state[rows]), f"shard {g} step {step}: states"
        assert np.array_equal(single._ctx.get_states(0, h * h), final_states[g])
        single.close()
