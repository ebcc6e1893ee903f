"""GPU tests of the caller side: FocusObserver and the DiscreteSteps-v0 harness."""

import os

import numpy as np
import pytest

from tests import helpers

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("pixels_per_thread", ["three", "automatic"])
def test_reference_notebook_outputs(pixels_per_thread, monkeypatch):
    """examples/environment.ipynb of the reference holds real outputs of the reference
    itself (numba on CUDA + OpenCV 4.9): DiscreteSteps(render_mode="rgb_array"),
    reset() -> [0.46703607, -0.84483975, 0, 0]; render(); step(8) ->
    ([0.59203607, -0.873161, 0.0625, -0.01416067], -1.4981610774993896, False, False)
    with final state [[5.311405, 8.98009]].  The initial state is recovered from those
    numbers (the reference's initializer is unseeded); everything else -- seed-0 RNG
    states, the 13-env extrema render, the 300 px render, the 600 px render that re-seeds,
    the next 300 px render, gray/median/Laplacian/var, normalisation -- must reproduce
    the printed digits exactly."""
    from reinfocus_amd.environments import harness

    if pixels_per_thread == "automatic":  # (one environment: the library's own choice is one pixel per thread)
        monkeypatch.delenv("REINFOCUS_RENDER_SETS", raising=False)
    env = harness.DiscreteSteps(render_mode="rgb_array", device=0)
    obs, _ = env.reset(state=[[5.311405, 8.66759]])
    assert obs.dtype == np.float32
    # the notebook shows numpy's 8-digit array repr: compare in that representation
    assert repr(obs) == "array([ 0.46703607, -0.84483975,  0.        ,  0.        ], dtype=float32)"
    image = env.render()  # HistoryVisualizer: 600 px rendering | 800 px wide plot
    assert image.shape == (600, 1400, 3) and image.dtype == np.uint8
    centre = image[250:350, 250:350]  # the (defocused) target: red / green checker, never blue
    assert np.all(centre[..., 2] == 0) and np.all(centre[..., 0].astype(int) + centre[..., 1] > 0)
    assert image[5, 5, 2] > 200  # sky in the corner
    obs, reward, terminated, truncated, _ = env.step(8)
    assert repr(obs) == "array([ 0.59203607, -0.873161  ,  0.0625    , -0.01416067], dtype=float32)"
    assert reward == -1.4981610774993896
    assert not terminated and not truncated
    assert np.array_equal(env._state, np.array([[5.311405, 8.98009]], dtype=np.float32))
    env.close()


def test_focus_observer_contract():
    """tests/environments/state_observer_test.py:424-480 of the reference."""
    from reinfocus_amd.environments import state_observer
    from reinfocus_amd.graphics import render

    num_envs = 5
    ends = (5, 10)
    renderer = render.FastRenderer(device=0)  # 100 spp, as the reference's test
    testee = state_observer.FocusObserver(num_envs, 0, 1, ends, renderer)  # 300 px
    assert np.all(testee.single_observation_space.low < testee.single_observation_space.high)
    assert testee.observation_space.shape == (num_envs, 1)

    # focus plane approaching the target from either side: focus value must rise
    for mid_point in np.linspace(*ends, num_envs):
        state = np.vstack([np.linspace(ends[0], mid_point, num_envs),
                           np.linspace(ends[1], mid_point, num_envs)]).T.astype(np.float32)
        fv = testee.observe(state)
        assert fv.shape == (num_envs, 1) and fv.dtype == np.float64
        gaps = np.abs(state[:, 0] - state[:, 1])
        order = np.argsort(-gaps, kind="stable")
        distinct = gaps[order][:-1] - gaps[order][1:] > 0.3
        assert np.all((fv[order, 0][1:] > fv[order, 0][:-1])[distinct])

    indices = np.array([True, False, True, False, True])
    part = testee.observe(np.array([[5, 10], [7.5, 10], [10, 10]], dtype=np.float32), indices)
    assert part.shape == (3, 1)
    assert np.all(part[1:] > part[:-1])
    part = testee.reset(np.array([[5, 10], [10, 10]], dtype=np.float32), np.array([True, False, False, False, True]))
    assert part.shape == (2, 1)


def test_vector_environment_steps_and_auto_resets(oracle, kernel_choice):
    from reinfocus_amd import registration

    n = 64
    env = registration.make_vec("DiscreteSteps-v0", num_envs=n, vectorization_mode="custom",
                                vector_kwargs={"frame_height": 32, "samples_per_pixel": 4, "seed": 3, "device": 0})
    obs, info = env.reset()
    assert obs.shape == (n, 4) and obs.dtype == np.float32 and info == {}
    assert np.all(obs[:, 2:] == 0) and np.all(np.abs(obs) <= 1)
    rng = np.random.default_rng(0)
    resets = 0
    for _ in range(30):
        actions = rng.integers(0, 13, n)
        before = env._state.copy()
        obs, rew, term, trunc, _ = env.step(actions)
        assert obs.shape == (n, 4) and rew.shape == (n,) and term.shape == (n,) and trunc.shape == (n,)
        assert not term.any()
        assert np.all(np.abs(obs) <= 1)
        done = term | trunc
        resets += int(done.sum())
        # envs that were not reset moved by the chosen step, clipped to [5, 10]
        moved = np.clip(before[:, 1].astype(np.float64) + env._action_set[actions], 5, 10).astype(np.float32)
        assert np.array_equal(env._state[~done, 1], moved[~done])
        assert np.all(obs[done, 2:] == 0)  # reset observations carry zero deltas
    assert resets > 0
    env.close()


# rf_env_step's schedules (rf_abi_env.hip).  Default for the canonical camera: the step's two renders and two focus
# measures as ONE launch each -- the environments that end are ranked before the render (the flags depend on their
# counters alone) and the blocks of the slots whose RNG streams the re-rendered frames continue make two passes --
# enqueued in one go ("fused") and replayed as one hipGraph from the second step on ("fused-graph").  With
# REINFOCUS_ENV_FUSED=0 the launches are separate: small configurations enqueue the whole step with one host
# synchronisation ("one-sync") and replay it as a graph ("graph"); large ones synchronise once mid-step and size the
# auto-reset launch by the count ("count-sized": rf_env_step_begin + rf_env_step_end, what a sharded environment
# uses).  The context reads the knobs at rf_create.
STEP_BRANCHES = {"fused-graph": {}, "fused": {"REINFOCUS_ENV_GRAPH": "0"},
                 # (the two-pass form of the kernel without cooperative tails: what launches of few blocks take)
                 "fused-graph, plain kernel": {"REINFOCUS_RENDER_COOP": "0"},
                 "graph": {"REINFOCUS_ENV_FUSED": "0"},
                 "one-sync": {"REINFOCUS_ENV_FUSED": "0", "REINFOCUS_ENV_GRAPH": "0"},
                 "count-sized": {"REINFOCUS_ENV_FUSED": "0", "REINFOCUS_ENV_ONE_SYNC_MAX": "0"}}
FIRST_STEP_BRANCH = {"fused-graph": "fused", "graph": "one-sync"}  # (a graph is captured from the second step on)
BRANCH_NAME = {"fused-graph, plain kernel": "fused-graph"}


@pytest.mark.parametrize("branch", list(STEP_BRANCHES))
@pytest.mark.parametrize("n,height,spp,steps", [(64, 32, 4, 45), (300, 16, 2, 30), (24, 132, 1, 40)])
def test_device_resident_step_equals_host_harness(n, height, spp, steps, branch, monkeypatch, kernel_choice):
    """rf_env_* (transformer, enders, scene packing, normaliser, rewards, auto-reset on the
    GPU) against the numpy harness: identical observations, rewards, flags and states for
    the same seeds and actions, step by step, including the partial auto-reset renders -- on
    every branch of rf_env_step (132-pixel frames: the strip kernel, with two passes, with a
    zero count, and with slots marked to be skipped)."""
    from reinfocus_amd.environments import harness

    kw = dict(num_envs=n, frame_height=height, samples_per_pixel=spp, seed=11, device=0)
    host = harness.VectorDiscreteSteps(**kw)
    for name, value in STEP_BRANCHES[branch].items():
        monkeypatch.setenv(name, value)
    dev = harness.DeviceVectorDiscreteSteps(**kw)
    for name in STEP_BRANCHES[branch]:
        monkeypatch.delenv(name)
    o_h, _ = host.reset()
    o_d, _ = dev.reset()
    assert o_d.dtype == np.float32 and np.array_equal(o_h, o_d)
    assert np.array_equal(host._state, dev._state)
    rng = np.random.default_rng(5)
    resets = 0
    for step in range(steps):
        actions = rng.integers(0, 13, n)
        oh, rh, th, ch, _info = host.step(actions)
        od, rd, td, cd, _info = dev.step(actions)
        assert np.array_equal(oh, od)
        assert rd.dtype == np.float64 and np.array_equal(rh, rd)
        assert np.array_equal(th, td) and np.array_equal(ch, cd)
        assert np.array_equal(host._state, dev._state)
        resets += int(ch.sum())
        # (a graph is captured from the second step on, when every buffer has its final size)
        name = BRANCH_NAME.get(branch, branch)
        assert dev._ctx.env_last_step_branch() == (FIRST_STEP_BRANCH.get(name, name) if step == 0 else name)
    assert resets > 0
    # (the library's own choice for launches of this size is the kernel without cooperative tails)
    plain = "plain" in branch or kernel_choice == "the library's choice"
    assert plain == dev._ctx.render_kernel_name().startswith("render_kernel<"), dev._ctx.render_kernel_name()
    # both initializers consumed the same number of draws
    assert host._initializer._generator.bit_generator.state == dev._initializer._generator.bit_generator.state
    host.close()
    dev.close()


def test_literal_drop_in_route_equals_the_resident_one(kernel_choice):
    """INTEGRATION.md's literal stub -- rf_render(host_out) -> numpy array -> rf_upload_frames + rf_focus on a scratch
    context (FastRenderer(host_frames=True), what `bench.py --env literal` measures) -- against frames that stay in
    HBM: same frames, same observations, rewards, flags and states through auto-resets."""
    from reinfocus_amd.environments import harness
    from reinfocus_amd.graphics import render

    kw = dict(num_envs=48, frame_height=24, samples_per_pixel=3, seed=7, device=0)
    resident = harness.VectorDiscreteSteps(**kw)
    literal = harness.VectorDiscreteSteps(host_frames=True, **kw)
    o_r, _ = resident.reset()
    o_l, _ = literal.reset()
    assert np.array_equal(o_r, o_l)
    rng = np.random.default_rng(3)
    resets = 0
    for _ in range(25):
        actions = rng.integers(0, 13, kw["num_envs"])
        a, b = resident.step(actions), literal.step(actions)
        for x, y in zip(a[:4], b[:4]):
            assert np.array_equal(x, y)
        assert np.array_equal(resident._state, literal._state)
        resets += int(a[3].sum())
    assert resets > 0
    frames = literal._renderer.render(24)
    assert isinstance(frames, np.ndarray) and frames.dtype == np.uint8 and frames.shape[1:] == (24, 24, 3)
    assert isinstance(resident._renderer.render(24), render.DeviceFrames)
    resident.close()
    literal.close()


def test_device_info_names_the_gpu():
    from reinfocus_amd import _native

    info = _native.device_info(0)
    bus = info["pci_bus_id"]
    assert info["device"] == 0 and len(bus.split(":")) == 3 and bus == bus.lower() and isinstance(info["numa_node"], int)
    with pytest.raises(AssertionError):
        _native.device_info(_native.device_count())


def test_env_step_graph_survives_other_calls_on_the_context(kernel_choice):
    """Small configurations replay rf_env_step as one hipGraph from their second step on.  The
    graph holds device pointers and kernel arguments by value, so every call that may reallocate a
    buffer or change the scene drops it; the steps around such calls must stay identical to the
    numpy harness (which never sees them)."""
    from reinfocus_amd import vision
    from reinfocus_amd.environments import harness

    kw = dict(num_envs=48, frame_height=32, samples_per_pixel=3, seed=4, device=0)
    host = harness.VectorDiscreteSteps(**kw)
    dev = harness.DeviceVectorDiscreteSteps(**kw)
    assert np.array_equal(host.reset()[0], dev.reset()[0])
    rng = np.random.default_rng(8)
    foreign = rng.integers(0, 256, size=(3, 80, 64, 3), dtype=np.uint8)
    for step in range(24):
        if step % 7 == 3:  # someone scores foreign frames on the environment's own context
            dev._ctx.upload_frames(foreign)
            values = dev._ctx.focus(3, 80, 64, vision.GRAY_MODE)
            assert values.shape == (3,) and np.all(values > 0)
        actions = rng.integers(0, 13, 48)
        want = host.step(actions)
        got = dev.step(actions)
        for a, b in zip(want[:4], got[:4]):
            assert np.array_equal(a, b)
        assert np.array_equal(host._state, dev._state)
    host.close()
    dev.close()


def test_continuous_jumps_on_gpu():
    from reinfocus_amd import registration

    env = registration.make("ContinuousJumps-v0", frame_height=64, samples_per_pixel=8, seed=2, device=0)
    obs, _ = env.reset()
    assert obs.shape == (4,) and np.all(np.abs(obs) <= 1)
    target = env._state[0, 0]
    # jump onto the target: the focus value rises; staying there earns the +1 bonus
    action = np.float32((target - 5.0) / 5.0 * 2.0 - 1.0)
    obs_on, reward_on, term, trunc, _ = env.step(action)
    obs_stay, reward_stay, *_ = env.step(action)
    assert not term and not trunc
    assert obs_on[1] > obs[1] or abs(env._state[0, 1] - target) < 1e-5
    assert reward_stay == np.float64(obs_stay[1]) + 1.0
    env.close()


def test_vector_env_visualizer_follows_auto_resets(kernel_choice):
    """vector_environment.py:137-158 with render_mode="rgb_array": the visualiser is reset for
    the done environments and stepped for the others; render() stacks one row per environment
    and, through its 600 px render, advances the RNG states like the reference does."""
    from reinfocus_amd.environments import harness

    num_envs = 3
    env = harness.VectorDiscreteSteps(max_episode_steps=2, num_envs=num_envs, render_mode="rgb_array",
                                      frame_height=64, samples_per_pixel=4, seed=5, device=0)
    twin = harness.VectorDiscreteSteps(max_episode_steps=2, num_envs=num_envs, frame_height=64,
                                       samples_per_pixel=4, seed=5, device=0)
    env.reset()
    twin.reset()
    image = env.render()
    assert image.shape == (num_envs * 600, 1400, 3)
    twin._renderer.render(600)  # what render() costs the RNG streams
    actions = np.array([6, 7, 5])
    for step in range(2):
        got = env.step(actions)
        want = twin.step(actions)
        for a, b in zip(got[:4], want[:4]):
            assert np.array_equal(a, b)
    assert got[3].all()  # time limit: every environment was reset in the second step
    visualizer = env._visualizer
    assert list(visualizer._current_moves) == [0, 0, 0]
    assert np.array_equal(visualizer._targets, env._state[:, 0])
    assert all(len(visualizer._move_histories.get_history(i)) == 1 for i in range(num_envs))
    env.close()
    twin.close()


def test_sb3_wrapper_over_device_environment():
    """vector_shim.SB3Wrapper over the device-resident environment: the stable-baselines3
    protocol returns exactly what the gymnasium-style API returns."""
    from reinfocus_amd.environments import harness, vector_shim

    kwargs = dict(max_episode_steps=3, num_envs=6, frame_height=64, samples_per_pixel=4, seed=11, device=0)
    wrapped = vector_shim.SB3Wrapper(harness.DeviceVectorDiscreteSteps(**kwargs), None)
    plain = harness.DeviceVectorDiscreteSteps(**kwargs)
    assert np.array_equal(wrapped.reset(), plain.reset()[0])
    generator = np.random.default_rng(2)
    ended = 0
    for _ in range(5):
        actions = generator.integers(0, 13, size=6)
        obs, rewards, dones, infos = wrapped.step(actions)
        want_obs, want_rewards, terminated, truncated, _ = plain.step(actions)
        assert np.array_equal(obs, want_obs) and np.array_equal(rewards, want_rewards)
        assert np.array_equal(dones, terminated | truncated)
        for i in range(6):
            assert ("terminal_observation" in infos[i]) == bool(dones[i])
        ended += int(dones.sum())
    assert ended >= 6  # the 3-step time limit fired for every environment
    assert wrapped.get_images() == [None]
    wrapped.close()
    plain.close()


def test_sb3_rollout_over_the_device_environment_equals_the_numpy_glue():
    """The adapter as stable-baselines3's on-policy rollout drives a VecEnv (vector_shim.py:63-93 under
    OnPolicyAlgorithm.collect_rollouts: reset once, then step_async / step_wait, the bootstrap value taken from
    infos[i]["terminal_observation"] wherever dones[i]) over the device-resident environment -- against the same adapter over the
    numpy-glue environment, an independent implementation of the step (transformer, enders, rewarder, normaliser in numpy around
    rf_render / rf_focus): observations, rewards, dones, terminal observations, the attribute plumbing and the rendered images."""
    from reinfocus_amd.environments import harness, vector_shim

    kwargs = dict(max_episode_steps=4, num_envs=5, frame_height=64, samples_per_pixel=4, seed=21, device=0, render_mode="rgb_array")
    device = vector_shim.SB3Wrapper(harness.DeviceVectorDiscreteSteps(**kwargs), "rgb_array")
    glue = vector_shim.SB3Wrapper(harness.VectorDiscreteSteps(**kwargs), "rgb_array")
    try:
        assert device.num_envs == glue.num_envs == 5
        assert device.get_attr("num_envs") == glue.get_attr("num_envs") == [5] * 5
        assert device.get_attr("num_envs", [0, 3]) == [5, 5] and device.env_is_wrapped(object) == [False] * 5
        last = [device.reset(), glue.reset()]
        assert np.array_equal(*last)
        generator = np.random.default_rng(8)
        bootstrapped = 0
        for step in range(9):
            actions = generator.integers(0, 13, size=5)
            device.step_async(actions)
            glue.step_async(actions)
            results = [device.step_wait(), glue.step_wait()]
            for got, want in zip(results[0][:3], results[1][:3]):
                assert got.dtype == want.dtype and np.array_equal(got, want), step
            for info_d, info_g, done in zip(results[0][3], results[1][3], results[0][2]):
                assert ("terminal_observation" in info_d) == ("terminal_observation" in info_g) == bool(done)
                if done:
                    assert np.array_equal(info_d["terminal_observation"], info_g["terminal_observation"])
                    bootstrapped += 1
            if step in (2, 7):  # (the visualiser's 600 px render re-seeds / advances the RNG states the next step uses: both alike)
                images = [device.get_images(), glue.get_images()]
                assert len(images[0]) == len(images[1]) == 1 and np.array_equal(images[0][0], images[1][0])
        assert bootstrapped >= 10
    finally:
        device.close()
        glue.close()


def test_registered_vector_env_is_device_resident_and_reproduces_the_notebook():
    """The env id's vector entry point (examples/__init__.py:6-11) builds the device-resident
    environment -- the one bench.py measures -- and, with render_mode="rgb_array", that environment
    prints the reference notebook's digits (see test_reference_notebook_outputs): rf_env_reset,
    the visualiser's 600 px render through rf_env_render (which re-seeds), rf_env_step."""
    from reinfocus_amd import registration
    from reinfocus_amd.environments import harness

    env = registration.make_vec("DiscreteSteps-v0", num_envs=1, vectorization_mode="custom",
                                vector_kwargs={"render_mode": "rgb_array", "device": 0})
    assert type(env) is harness.DeviceVectorDiscreteSteps and env.render_mode == "rgb_array"
    obs, _ = env.reset(state=[[5.311405, 8.66759]])
    assert repr(obs[0]) == "array([ 0.46703607, -0.84483975,  0.        ,  0.        ], dtype=float32)"
    image = env.render()
    assert image.shape == (600, 1400, 3) and image.dtype == np.uint8
    centre = image[250:350, 250:350]
    assert np.all(centre[..., 2] == 0) and np.all(centre[..., 0].astype(int) + centre[..., 1] > 0)
    obs, reward, terminated, truncated, _ = env.step(np.array([8]))
    assert repr(obs[0]) == "array([ 0.59203607, -0.873161  ,  0.0625    , -0.01416067], dtype=float32)"
    assert reward[0] == -1.4981610774993896
    assert not terminated[0] and not truncated[0]
    assert np.array_equal(env._state, np.array([[5.311405, 8.98009]], dtype=np.float32))
    env.close()
    # the numpy-glue environment stays reachable under the same id
    host = registration.make_vec("DiscreteSteps-v0", num_envs=2, glue="host", frame_height=16, samples_per_pixel=1,
                                 device=0)
    assert type(host) is harness.VectorDiscreteSteps
    host.close()


def test_device_environment_visualiser_equals_host_glue(kernel_choice):
    """render_mode="rgb_array" on the device-resident environment: same frames from render(), same
    observations afterwards (the 600 px render advances / re-seeds the RNG states), same rows after
    a partial auto-reset (the shared renderer then holds only the environments that were reset,
    vector_environment.py:144 + episode_visualizer.py:197-201) as the numpy-glue environment."""
    from reinfocus_amd.environments import harness

    kw = dict(max_episode_steps=5, num_envs=4, render_mode="rgb_array", frame_height=32, samples_per_pixel=2,
              seed=9, device=0)
    host = harness.VectorDiscreteSteps(**kw)
    dev = harness.DeviceVectorDiscreteSteps(**kw)
    start = [[7.5, 7.5]] * 4
    assert np.array_equal(host.reset(state=start)[0], dev.reset(state=start)[0])
    rows_seen = set()
    for step in range(8):
        a, b = host.render(), dev.render()
        assert a.shape == b.shape and a.shape[0] % 600 == 0
        assert np.array_equal(a[:, :600], b[:, :600])  # the rendered frames (plots are matplotlib's)
        rows_seen.add(a.shape[0] // 600)
        # environment 0 walks away from its target (+0.625 per step: ends after 3 diverging steps, alone);
        # the others stay put and end together at the time limit
        actions = np.array([9, 6, 6, 6])
        want, got = host.step(actions), dev.step(actions)
        for x, y in zip(want[:4], got[:4]):
            assert np.array_equal(x, y)
        assert np.array_equal(host._state, dev._state)
        for i in range(4):
            assert host._ender.status(i) == dev._shard.status(i)
        assert np.array_equal(host._visualizer._current_moves, dev._visualizer._current_moves)
        assert np.array_equal(host._visualizer._targets, dev._visualizer._targets)
    assert {1, 4} <= rows_seen  # full sets and the one-row set after environment 0's lone auto-reset
    host.close()
    dev.close()


def _toward_target(state, action_set):
    """An action per environment that never grows |target - focus| (nobody diverges)."""
    gap = state[:, 0] - state[:, 1]
    step = 0.15625
    return np.where(gap > step, 7, np.where(gap < -step, 5, 6)).astype(np.int64)


@pytest.mark.parametrize("n,shards", [(12, 2), (7, 3)])
def test_sharded_environment_equals_one_device(n, shards, kernel_choice):
    """harness.ShardedVectorDiscreteSteps with several contexts on device 0 against ONE
    DeviceVectorDiscreteSteps holding all environments: bit-identical observations, rewards and
    states while renders are full ones (global RNG-state indices via rf_seed's first_state_index);
    after auto-resets (documented deviation: partial renders index their states per shard) the
    initializer's states still go to the ended environments in GLOBAL index order."""
    from reinfocus_amd.environments import harness

    kw = dict(max_episode_steps=6, num_envs=n, frame_height=24, samples_per_pixel=3, seed=21)
    one = harness.DeviceVectorDiscreteSteps(device=0, **kw)
    many = harness.ShardedVectorDiscreteSteps(devices=[0] * shards, **kw)
    assert [c for _, c in many._ranges] == [n // shards + (1 if g < n % shards else 0) for g in range(shards)]
    o1, _ = one.reset()
    o2, _ = many.reset()
    assert o2.dtype == np.float32 and np.array_equal(o1, o2)
    assert np.array_equal(one._state, many._state)
    for _ in range(5):  # full renders only: nobody diverges, the time limit is at step 6
        actions = _toward_target(one._state, one._action_set)
        a, b = one.step(actions), many.step(actions)
        for x, y in zip(a[:4], b[:4]):
            assert np.array_equal(x, y)
        assert not b[3].any() and np.array_equal(one._state, many._state)
    # step 6: every environment ends; both draw the same n states for them, in index order
    a, b = one.step(np.full(n, 6)), many.step(np.full(n, 6))
    assert a[3].all() and b[3].all() and np.array_equal(a[1], b[1])
    assert np.array_equal(one._state, many._state)
    assert one._initializer._generator.bit_generator.state == many._initializer._generator.bit_generator.state
    # from here on the two are different sample paths; the sharded one keeps its invariants, and a
    # twin generator predicts the states its ended environments receive
    twin = harness._Initializer((5.0, 10.0), 21)
    twin.initialize(n)
    twin.initialize(n)
    rng = np.random.default_rng(3)
    ended = 0
    for _ in range(8):
        actions = rng.integers(0, 13, n)
        obs, rewards, terminated, truncated, _ = many.step(actions)
        assert obs.shape == (n, 4) and np.all(np.abs(obs) <= 1) and not terminated.any()
        k = int(truncated.sum())
        if k:
            assert np.array_equal(many._state[truncated], twin.initialize(k))
            assert np.all(obs[truncated, 2:] == 0)
        ended += k
    assert ended > 0
    one.close()
    many.close()


@pytest.mark.parametrize("shards", [2, 3])
def test_sharded_exact_mode_equals_one_device_through_auto_resets(shards, kernel_choice):
    """exact=True (rf_env_render_states / rf_env_step_end_given): the compacted row r of an auto-reset
    is rendered by the shard owning environment slot r, as on one device (vector_environment.py:144 ->
    state_observer.py:377-381 -> render.py:217) -- so 2 and 3 shards reproduce ONE
    DeviceVectorDiscreteSteps bit for bit, observations, rewards, flags, environment states and RNG
    states, through many auto-resets of few and of all environments."""
    from reinfocus_amd.environments import harness

    n, h = 13, 24
    kw = dict(max_episode_steps=5, num_envs=n, frame_height=h, samples_per_pixel=3, seed=31)
    one = harness.DeviceVectorDiscreteSteps(device=0, **kw)
    many = harness.ShardedVectorDiscreteSteps(devices=[0] * shards, exact=True, **kw)
    o1, _ = one.reset()
    o2, _ = many.reset()
    assert np.array_equal(o1, o2)
    rng = np.random.default_rng(77)
    steps_with_resets, sizes = 0, set()
    for _ in range(28):
        actions = rng.integers(0, 13, n)
        a, b = one.step(actions), many.step(actions)
        for x, y in zip(a[:4], b[:4]):
            assert x.dtype == y.dtype and np.array_equal(x, y)
        assert np.array_equal(one._state, many._state)
        k = int(b[3].sum())
        steps_with_resets += k > 0
        sizes.add(k)
    assert steps_with_resets >= 6 and len(sizes) >= 4  # partial sets of several sizes, and full ones
    assert one._initializer._generator.bit_generator.state == many._initializer._generator.bit_generator.state
    states = np.concatenate(many._each(lambda shard: shard.ctx.get_states()))
    assert np.array_equal(states, one._ctx.get_states())
    many.close()
    one.close()


def test_sharded_environment_renders_like_its_shards(kernel_choice):
    """render_mode="rgb_array" on the sharded environment: every shard draws what a single-context
    environment over its range draws (600 px frames from its own renderer state + the plots), stacked
    in shard order (vector_environment.py:166-176 -> episode_visualizer.py:188-201)."""
    from reinfocus_amd.environments import harness

    n, h = 3, 32
    kw = dict(max_episode_steps=20, frame_height=h, samples_per_pixel=2)
    many = harness.ShardedVectorDiscreteSteps(num_envs=n, devices=[0, 0], render_mode="rgb_array", seed=2, **kw)
    many.reset()
    state0 = many._state
    actions = np.array([5, 6, 7])
    many.step(actions)
    image = many.render()
    assert image.dtype == np.uint8 and image.shape[0] == n * 600 and image.shape[1] > 600
    many.close()
    top = 0
    for first, count in many._ranges:
        single = harness.DeviceVectorDiscreteSteps(num_envs=count, render_mode="rgb_array", seed=0, device=0,
                                                   first_state_index=first * h * h, **kw)
        single.reset(state=state0[first:first + count])
        single.step(actions[first:first + count])
        part = single.render()
        assert np.array_equal(image[top:top + count * 600], part)
        top += count * 600
        single.close()


def test_two_phase_step_guards(kernel_choice):
    """rf_env_step_begin / rf_env_step_end must alternate; rf_env_step refuses an open step."""
    from reinfocus_amd.environments import harness

    env = harness.DeviceVectorDiscreteSteps(num_envs=3, frame_height=16, samples_per_pixel=1, seed=0, device=0)
    env.reset()
    ctx = env._ctx
    with pytest.raises(AssertionError):
        ctx.env_step_end(np.zeros((0, 2), dtype=np.float32))
    rewards, truncated, k = ctx.env_step_begin(np.full(3, 6))
    assert k == 0 and not truncated.any() and rewards.shape == (3,)
    with pytest.raises(AssertionError):
        ctx.env_step_begin(np.full(3, 6))
    with pytest.raises(AssertionError):
        ctx.env_step(np.full(3, 6), np.zeros((3, 2), dtype=np.float32))
    obs = ctx.env_step_end(np.zeros((0, 2), dtype=np.float32))
    assert obs.shape == (3, 4)
    env.step(np.full(3, 6))
    env.close()


@pytest.mark.parametrize("fused", ["1", "0"])
def test_planned_step_equals_the_whole_step(fused, monkeypatch, kernel_choice):
    """rf_env_step_plan + rf_env_step_run (the halves a sharded environment uses: the cut is before the render) against
    rf_env_step with the same pool, through auto-resets, with and without the two-pass render kernel; and the guards:
    a planned step is finished by rf_env_step_run only."""
    from reinfocus_amd.environments import harness

    n = 96
    kw = dict(max_episode_steps=5, num_envs=n, frame_height=32, samples_per_pixel=3, seed=19, device=0)
    monkeypatch.setenv("REINFOCUS_ENV_FUSED", fused)
    whole, halves = harness.DeviceVectorDiscreteSteps(**kw), harness.DeviceVectorDiscreteSteps(**kw)
    monkeypatch.delenv("REINFOCUS_ENV_FUSED")
    assert np.array_equal(whole.reset()[0], halves.reset()[0])
    rng = np.random.default_rng(2)
    pools = np.random.default_rng(77)
    ended = 0
    for step in range(14):
        actions = rng.integers(0, 13, n)
        pool = pools.uniform(5, 10, size=(n, 2)).astype(np.float32)
        obs, rewards, truncated, used = whole._ctx.env_step(actions, pool)
        k = halves._ctx.env_step_plan(actions)
        assert k == used
        if step == 3:
            with pytest.raises(AssertionError):
                halves._ctx.env_step(actions, pool)
            with pytest.raises(AssertionError):
                halves._ctx.env_step_end(pool[:k])
            with pytest.raises(AssertionError):
                halves._ctx.env_step_plan(actions)
        got = halves._ctx.env_step_run(pool[:k])
        assert np.array_equal(got[0], obs) and np.array_equal(got[1], rewards) and np.array_equal(got[2], truncated)
        assert np.array_equal(whole._state, halves._state)
        assert halves._ctx.env_scene_len() == (k or n)
        ended += k
    assert ended > n
    assert np.array_equal(whole._ctx.get_states(), halves._ctx.get_states())
    rows = k or n  # (the frame buffer's leading rows hold the scene set rendered last on every schedule)
    frames = [env._ctx.get_frames((n, 32, 32), 0, rows) for env in (whole, halves)]
    assert np.array_equal(*frames)
    with pytest.raises(AssertionError):
        halves._ctx.env_step_run(np.zeros((0, 2), dtype=np.float32))
    halves._ctx.env_step_plan(rng.integers(0, 13, n))
    halves._ctx.env_step_abort()
    with pytest.raises(AssertionError):
        halves._ctx.env_step_plan(rng.integers(0, 13, n))  # (a dropped step: reset first)
    halves.reset()
    halves.step(rng.integers(0, 13, n))
    whole.close()
    halves.close()


@pytest.mark.parametrize("build,height", [("libreinfocus_skew.so", 128), ("libreinfocus_skew.so", 100),
                                          ("libreinfocus_cap32.so", 128), ("libreinfocus_skew.so", 132),
                                          ("libreinfocus_cap32.so", 164)])  # (132, 164: the strip kernel)
def test_fused_step_on_the_test_builds_of_the_library(build, height, tmp_path):
    """The two-pass instance of the render kernel (the fused step's: a block renders its tile, then the tile of the
    environment that takes its slot in the auto-reset's compacted set) under the test builds that
    tests/test_gpu_parity.py runs the single-pass instance under -- delayed waves (RF_TEST_SKEW: the passes are
    ordered against each other by barriers alone) and tiny cooperative lists (overflow paths in both passes) --, in a
    child process, against the shipped library's run of the same steps (which the tests above pin to the numpy glue)."""
    import subprocess
    import sys

    from reinfocus_amd.environments import harness

    so = helpers.built("tests/gpucheck", build)
    n, steps = 12, 7
    kw = dict(max_episode_steps=3, num_envs=n, frame_height=height, samples_per_pixel=4, seed=29, device=0)
    actions = np.random.default_rng(3).integers(0, 13, (steps, n))
    np.save(tmp_path / "actions.npy", actions)
    script = (
        "import sys; sys.path.insert(0, %r)\n"
        "import numpy as np\n"
        "from reinfocus_amd.environments import harness\n"
        "env = harness.DeviceVectorDiscreteSteps(**%r)\n"
        "out = [env.reset()[0]]\n"
        "for a in np.load(%r):\n"
        "    o, r, _, t, _ = env.step(a)\n"
        "    out += [o, r, t]\n"
        "assert env._ctx.env_last_step_branch() == 'fused-graph' and (env._ctx.render_kernel_name().endswith(', true>') or 'strip' in env._ctx.render_kernel_name())\n"
        "np.savez(%r, *out, states=env._ctx.get_states(0, 3 * %d))\n"
    ) % (helpers.ROOT, kw, str(tmp_path / "actions.npy"), str(tmp_path / "out.npz"), height * height)
    subprocess.check_call([sys.executable, "-c", script], env=dict(os.environ, REINFOCUS_HIP_LIB=so))
    got = np.load(tmp_path / "out.npz")
    env = harness.DeviceVectorDiscreteSteps(**kw)
    want = [env.reset()[0]]
    ended = 0
    for a in actions:
        o, r, _, t, _ = env.step(a)
        want += [o, r, t]
        ended += int(t.sum())
    assert ended > n
    for i, w in enumerate(want):
        assert np.array_equal(got[f"arr_{i}"], w), i
    assert np.array_equal(got["states"], env._ctx.get_states(0, 3 * height * height))
    env.close()


def test_small_environments_take_the_one_pixel_kernel(monkeypatch):
    """The library's own choice for launches of few blocks (rf_abi_render.hip few_blocks: the reference's default single
    environment among them): the kernel without cooperative tails, in its two-pass form inside the fused step -- a
    device-resident environment of 3 x 64 x 64 at 8 samples, equal to the numpy glue."""
    from reinfocus_amd.environments import harness

    monkeypatch.delenv("REINFOCUS_RENDER_SETS", raising=False)
    kw = dict(max_episode_steps=4, num_envs=3, frame_height=64, samples_per_pixel=8, seed=5, device=0)
    host, dev = harness.VectorDiscreteSteps(**kw), harness.DeviceVectorDiscreteSteps(**kw)
    assert np.array_equal(host.reset()[0], dev.reset()[0])
    rng = np.random.default_rng(4)
    for step in range(11):
        actions = rng.integers(0, 13, 3)
        for x, y in zip(host.step(actions)[:4], dev.step(actions)[:4]):
            assert np.array_equal(x, y)
        assert dev._ctx.env_last_step_branch() == ("fused" if step == 0 else "fused-graph")
        assert dev._ctx.render_kernel_name() == "render_kernel<true, true, true>"  # (its two-pass form)
    assert np.array_equal(host._state, dev._state)
    host.close()
    dev.close()


def test_env_step_graph_capture_failure_falls_back(monkeypatch, kernel_choice):
    """rf_env_step's hipGraph branch when instantiation fails (REINFOCUS_ENV_GRAPH_FAIL=1 makes the
    first one fail): the step is enqueued call by call from then on, with identical results."""
    from reinfocus_amd.environments import harness

    kw = dict(num_envs=40, frame_height=32, samples_per_pixel=2, seed=6, device=0)
    host = harness.VectorDiscreteSteps(**kw)
    monkeypatch.setenv("REINFOCUS_ENV_GRAPH_FAIL", "1")
    dev = harness.DeviceVectorDiscreteSteps(**kw)
    monkeypatch.delenv("REINFOCUS_ENV_GRAPH_FAIL")
    assert np.array_equal(host.reset()[0], dev.reset()[0])
    rng = np.random.default_rng(12)
    for _ in range(12):
        actions = rng.integers(0, 13, 40)
        for x, y in zip(host.step(actions)[:4], dev.step(actions)[:4]):
            assert np.array_equal(x, y)
    host.close()
    dev.close()
