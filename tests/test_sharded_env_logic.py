"""Host logic of harness.ShardedVectorDiscreteSteps and of the gymnasium base classes, on the CPU.

The shards' rf_ctx is replaced by a numpy stand-in with the two-phase step protocols
(env_step_plan / env_step_run; env_step_begin / env_step_end_given in the exact mode); what is under test is what the sharded environment adds:
contiguous env ranges, global RNG-state offsets, one thread per shard, the initializer's rows
handed out in GLOBAL index order across shards, host concatenation.  The real thing runs in
tests/test_gpu_environment.py::test_sharded_environment_equals_one_device."""

import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class FakeContext:
    """The part of _native.Context a shard uses, in numpy: an environment ends every
    `period[e]` steps; observations encode (state, steps) so that mix-ups are visible."""

    def __init__(self, n, first_env):
        self.n = n
        self.period = 2 + (np.arange(first_env, first_env + n) % 3)
        self.state = np.zeros((n, 2), dtype=np.float32)
        self.steps = np.zeros(n, dtype=np.int64)
        self.pending = None
        self.closed = False
        self.first_env = first_env
        self.rendered_rows = 0
        self.aborted = False
        self.fail_next_begin = False
        self.fail_next_end = False
        self.last_k = None  # rows of the scene set uploaded last: k after a partial auto-reset, else all (rf_env_scene_len)

    def _obs(self):
        return np.column_stack([self.state, self.steps, self.steps * 0]).astype(np.float32)

    def env_reset(self, states):
        self.state = np.array(states, dtype=np.float32).reshape(self.n, 2)
        self.steps[:] = 0
        self.last_k = None
        return self._obs()

    def env_step_begin(self, actions):
        assert self.pending is None
        if self.fail_next_begin:
            self.fail_next_begin = False
            raise RuntimeError("injected failure")
        self.state[:, 1] += np.asarray(actions, dtype=np.float32) * 0.125
        self.steps += 1
        truncated = self.steps >= self.period
        self.pending = truncated
        return self.state.sum(axis=1).astype(np.float64), truncated.copy(), int(truncated.sum())

    def env_step_end(self, rows):
        if self.fail_next_end:
            self.fail_next_end = False
            raise RuntimeError("injected failure in the second half")
        truncated, self.pending = self.pending, None
        rows = np.asarray(rows, dtype=np.float32).reshape(-1, 2)
        assert len(rows) == truncated.sum()
        self.last_k = int(truncated.sum()) or None
        self.state[truncated] = rows
        self.steps[truncated] = 0
        return self._obs()

    def env_step_plan(self, actions):  # the default mode's halves: the cut is BEFORE the render
        return self.env_step_begin(actions)[2]

    def env_step_run(self, rows):
        truncated = self.pending.copy()
        rewards = self.state.sum(axis=1).astype(np.float64)  # (before the rows are applied, as env_step_begin reports them)
        return self.env_step_end(rows), rewards, truncated

    def env_render_states(self, states):
        """Exact mode: "focus values" that say where a row was rendered (slot = first env + local row)."""
        states = np.asarray(states, dtype=np.float32).reshape(-1, 2)
        assert 0 < len(states) <= self.n
        self.rendered_rows += len(states)
        return (self.first_env + np.arange(len(states))) * 100.0 + states[:, 0].astype(np.float64)

    def env_step_end_given(self, rows, focus):
        truncated, self.pending = self.pending, None
        rows = np.asarray(rows, dtype=np.float32).reshape(-1, 2)
        assert len(rows) == truncated.sum() == len(focus)
        self.state[truncated] = rows
        self.steps[truncated] = 0
        obs = self._obs()
        obs[truncated, 3] = np.asarray(focus, dtype=np.float32)
        return obs

    def env_step_abort(self):
        self.aborted = self.pending is not None
        self.pending = None

    def env_states(self):
        return self.state.copy()

    def close(self):
        self.closed = True


@pytest.fixture()
def fake_shards(monkeypatch):
    from reinfocus_amd import _native
    from reinfocus_amd.environments import harness

    class Made(list):
        pins = None

    made = Made()

    class FakeShard:
        ENDS, TARGET_RADIUS, MAX_MOVE = harness._DeviceShard.ENDS, 0.25, 5.0
        action_set = np.arange(13, dtype=np.float64)

        def __init__(self, num_envs, max_episode_steps, frame_height, samples_per_pixel, device, first_state_index):
            assert first_state_index % (frame_height * frame_height) == 0
            self.first_env = first_state_index // (frame_height * frame_height)
            self.num_envs, self.device = num_envs, device
            self.ctx = FakeContext(num_envs, self.first_env)
            made.append(self)

        def render(self, frame_height):  # what HistoryVisualizer asks of a renderer: the scene set uploaded last
            rows = self.ctx.last_k or self.num_envs
            return np.full((rows, frame_height, frame_height, 3), self.first_env, dtype=np.uint8)

        def status(self, index):  # ... and of an ender
            return f"env {self.first_env + index}"

    monkeypatch.setattr(harness, "_DeviceShard", FakeShard)
    monkeypatch.setattr(_native, "device_count", lambda: 4)
    # a host with two NUMA nodes: devices 0, 1 on node 0 (CPUs 0-3), devices 2, 3 on node 1 (CPUs 4-7)
    monkeypatch.setattr(_native, "device_info",
                        lambda d: {"device": d, "pci_bus_id": f"0000:{0x10 + d:02x}:00.0", "numa_node": d // 2})
    pins = []

    def fake_pin(node, whole_process=False, sysfs=None):
        import threading

        pins.append((threading.current_thread().name, node))
        return list(range(4 * node, 4 * node + 4))

    monkeypatch.setattr(_native, "pin_to_numa_node", fake_pin)
    made.pins = pins
    return made


def test_shard_threads_pin_themselves_to_their_gpus_numa_node(fake_shards):
    from reinfocus_amd.environments import harness

    env = harness.ShardedVectorDiscreteSteps(num_envs=8, devices=[0, 1, 2, 3], frame_height=16, samples_per_pixel=1, seed=1)
    assert [p["pci_bus_id"] for p in env.placements] == ["0000:10:00.0", "0000:11:00.0", "0000:12:00.0", "0000:13:00.0"]
    assert [p["numa_node"] for p in env.placements] == [0, 0, 1, 1]
    assert [p["cpus"] for p in env.placements] == [[0, 1, 2, 3], [0, 1, 2, 3], [4, 5, 6, 7], [4, 5, 6, 7]]
    # every shard pinned its OWN thread (the one its context is driven from), not the caller's
    assert sorted(fake_shards.pins) == [(f"reinfocus-shard{g}_0", g // 2) for g in range(4)]
    env.close()
    del fake_shards.pins[:]
    env = harness.ShardedVectorDiscreteSteps(num_envs=4, devices=[0, 3], frame_height=16, samples_per_pixel=1, seed=1,
                                             numa_pin=False)
    assert fake_shards.pins == [] and [p["cpus"] for p in env.placements] == [None, None]
    env.close()


def test_numa_helpers(tmp_path):
    from reinfocus_amd import _native

    (tmp_path / "node1").mkdir()
    (tmp_path / "node1" / "cpulist").write_text("2-3,9,64-66\n")
    assert _native.numa_cpus(1, sysfs=str(tmp_path)) == {2, 3, 9, 64, 65, 66}
    assert _native.numa_cpus(5, sysfs=str(tmp_path)) == set()
    assert _native.pin_to_numa_node(-1) is None and _native.pin_to_numa_node(None) is None
    # a node none of whose CPUs this process may use changes nothing
    (tmp_path / "node2").mkdir()
    (tmp_path / "node2" / "cpulist").write_text("100000-100001\n")
    assert _native.pin_to_numa_node(2, sysfs=str(tmp_path)) is None
    # a node that contains this thread's current CPUs: pinned to the intersection, and harmless
    import os

    now = sorted(os.sched_getaffinity(0))
    (tmp_path / "node3").mkdir()
    (tmp_path / "node3" / "cpulist").write_text(",".join(str(c) for c in now) + ",100000\n")
    assert _native.pin_to_numa_node(3, sysfs=str(tmp_path)) == now
    assert sorted(os.sched_getaffinity(0)) == now


def test_split_environments():
    from reinfocus_amd.environments import harness

    assert harness.split_environments(10, 3) == [(0, 4), (4, 3), (7, 3)]
    assert harness.split_environments(32768, 8) == [(g * 4096, 4096) for g in range(8)]  # BASELINE configs[3]
    assert harness.split_environments(3, 3) == [(0, 1), (1, 1), (2, 1)]


@pytest.mark.parametrize("n,devices", [(11, [0, 1, 2]), (8, [0, 1, 2, 3]), (5, [3])])
def test_sharded_step_equals_one_shard(fake_shards, n, devices):
    from reinfocus_amd.environments import harness

    many = harness.ShardedVectorDiscreteSteps(num_envs=n, devices=devices, frame_height=16, samples_per_pixel=1, seed=4)
    shards = sorted(fake_shards, key=lambda s: s.first_env)  # built on the pool's threads, in any order
    assert [s.device for s in shards] == devices
    assert [(s.first_env, s.num_envs) for s in shards] == harness.split_environments(n, len(devices))
    one = harness.ShardedVectorDiscreteSteps(num_envs=n, devices=[0], frame_height=16, samples_per_pixel=1, seed=4)
    assert many.observation_space.shape == (n, 4) and many.single_action_space.n == 13
    o1, _ = one.reset()
    o2, info = many.reset()
    assert info == {} and np.array_equal(o1, o2)
    rng = np.random.default_rng(0)
    total = 0
    for _ in range(12):
        actions = rng.integers(0, 13, n)
        a, b = one.step(actions), many.step(actions)
        for x, y in zip(a[:4], b[:4]):
            assert x.shape == y.shape and np.array_equal(x, y)
        assert np.array_equal(one._state, many._state)
        total += int(b[3].sum())
    assert total > n  # every environment ended at least once, in different steps
    # same number of initializer draws: the rows went out in global index order
    assert one._initializer._generator.bit_generator.state == many._initializer._generator.bit_generator.state
    many.close()
    one.close()
    assert all(s.ctx.closed for s in fake_shards)


def test_sharded_defaults_and_guards(fake_shards):
    from reinfocus_amd import registration
    from reinfocus_amd.environments import harness

    env = harness.ShardedVectorDiscreteSteps(num_envs=8, frame_height=8, samples_per_pixel=1)
    assert env.devices == [0, 1, 2, 3]  # every visible device by default
    env.close()
    with pytest.raises(AssertionError):
        harness.ShardedVectorDiscreteSteps(num_envs=2, devices=[0, 1, 2], frame_height=8)
    with pytest.raises(AssertionError):
        harness.ShardedVectorDiscreteSteps(num_envs=4, devices=[0], render_mode="human", frame_height=8)
    # the env id reaches it through `devices`
    env = registration.make_vec("DiscreteSteps-v0", num_envs=6, devices=[0, 1], frame_height=8, samples_per_pixel=1)
    assert type(env) is harness.ShardedVectorDiscreteSteps and env.num_envs == 6
    env.close()


@pytest.mark.parametrize("n,devices", [(11, [0, 1, 2]), (9, [0, 1, 2, 3])])
def test_exact_mode_renders_row_r_where_slot_r_lives(fake_shards, n, devices):
    """exact=True: the compacted row r of an auto-reset is rendered by the shard that owns environment
    slot r (its RNG states are the ones a single device would draw from, render.py:217) and the value
    comes back to the shard the environment lives on -- so several shards equal one."""
    from reinfocus_amd.environments import harness

    kw = dict(num_envs=n, frame_height=16, samples_per_pixel=1, seed=9, exact=True)
    many = harness.ShardedVectorDiscreteSteps(devices=devices, **kw)
    mine = list(fake_shards)
    one = harness.ShardedVectorDiscreteSteps(devices=[0], **kw)
    assert np.array_equal(one.reset()[0], many.reset()[0])
    rng = np.random.default_rng(2)
    total = 0
    for _ in range(10):
        actions = rng.integers(0, 13, n)
        a, b = one.step(actions), many.step(actions)
        for x, y in zip(a[:4], b[:4]):
            assert np.array_equal(x, y)
        k = int(b[3].sum())
        if k:  # the "focus values" name the slots 0..k-1 in order, whichever environments ended
            assert np.array_equal(b[0][b[3], 3] // 100, np.arange(k))
        total += k
    assert total > n
    # the resets were rendered by the first shards only
    rendered = [s.ctx.rendered_rows for s in sorted(mine, key=lambda s: s.first_env)]
    assert sum(rendered) == total and rendered[0] > 0 and rendered[-1] == 0
    many.close()
    one.close()


def test_a_failing_shard_does_not_leave_the_others_half_way(fake_shards):
    from reinfocus_amd.environments import harness

    env = harness.ShardedVectorDiscreteSteps(num_envs=9, devices=[0, 1, 2], frame_height=8, samples_per_pixel=1, seed=1)
    env.reset()
    shards = sorted(fake_shards, key=lambda s: s.first_env)
    with pytest.raises(AssertionError):  # checked for all shards before any of them begins
        env.step(np.array([0, 1, 2, 3, 4, 5, 6, 7, 13]))
    assert all(s.ctx.pending is None and not s.ctx.aborted for s in shards)
    shards[1].ctx.fail_next_begin = True
    with pytest.raises(RuntimeError, match="injected failure"):
        env.step(np.zeros(9, dtype=np.int64))
    assert shards[0].ctx.aborted and shards[2].ctx.aborted and not shards[1].ctx.aborted
    assert all(s.ctx.pending is None for s in shards)
    env.reset()
    env.step(np.zeros(9, dtype=np.int64))  # usable again after a reset
    env.close()


def test_a_failure_in_the_second_half_leaves_no_open_step(fake_shards):
    from reinfocus_amd.environments import harness

    env = harness.ShardedVectorDiscreteSteps(num_envs=9, devices=[0, 1, 2], frame_height=8, samples_per_pixel=1, seed=1)
    env.reset()
    shards = sorted(fake_shards, key=lambda s: s.first_env)
    shards[1].ctx.fail_next_end = True
    with pytest.raises(RuntimeError, match="second half"):
        env.step(np.zeros(9, dtype=np.int64))
    # the shard that failed still had its step open and dropped it; the others had finished theirs
    assert shards[1].ctx.aborted and all(s.ctx.pending is None for s in shards)
    env.reset()
    env.step(np.zeros(9, dtype=np.int64))
    env.close()


def test_renders_after_a_partial_reset_are_the_compacted_rows(fake_shards):
    """After a step in which k environments ended one device holds the k compacted rows (rf_env_scene_len); sharded,
    those are the shards that had resets, in shard order -- a shard without resets contributes nothing."""
    from reinfocus_amd.environments import harness

    env = harness.ShardedVectorDiscreteSteps(num_envs=6, devices=[0, 1], render_mode="rgb_array", frame_height=8,
                                             samples_per_pixel=1, seed=0)
    env.reset()
    assert env.render_frames().shape[0] == 6  # after a reset: every environment
    shards = sorted(fake_shards, key=lambda s: s.first_env)
    shards[0].ctx.period[:] = 100  # only shard 1's environments end in the next steps
    shards[1].ctx.period[:] = [1, 100, 1]
    *_, truncated, _ = env.step(np.zeros(6, dtype=np.int64))
    assert list(truncated) == [False, False, False, True, False, True]
    frames = env.render_frames()
    assert frames.shape[0] == 2 and set(frames[:, 0, 0, 0]) == {3}  # two compacted rows, both drawn by shard 1
    shards[1].ctx.period[:] = 100
    env.step(np.zeros(6, dtype=np.int64))  # nobody ended: the full sets again
    assert env.render_frames().shape[0] == 6
    env.close()
    with pytest.raises(AssertionError, match="exact"):
        harness.ShardedVectorDiscreteSteps(num_envs=4, devices=[0, 1], render_mode="rgb_array", exact=True, frame_height=8,
                                           samples_per_pixel=1)


def test_every_shard_has_its_own_thread(fake_shards):
    import threading

    from reinfocus_amd.environments import harness

    env = harness.ShardedVectorDiscreteSteps(num_envs=6, devices=[0, 1, 2], frame_height=8, samples_per_pixel=1)
    seen = [set() for _ in range(3)]
    for _ in range(5):
        names = env._each(lambda shard: threading.current_thread().name)
        for g, name in enumerate(names):
            seen[g].add(name)
    assert all(len(names) == 1 for names in seen) and len(set.union(*seen)) == 3
    env.close()


def test_sharded_render_mode_stacks_the_shards(fake_shards):
    from reinfocus_amd.environments import harness

    env = harness.ShardedVectorDiscreteSteps(num_envs=3, devices=[0, 1], render_mode="rgb_array", frame_height=8,
                                             samples_per_pixel=1, seed=0)
    env.reset()
    env.step(np.array([1, 2, 3]))
    frames = env.render_frames()
    assert frames.shape == (3, 600, 600, 3) and list(frames[:, 0, 0, 0]) == [0, 0, 2]  # shard 0: envs 0-1, shard 1: env 2
    image = env.render()
    assert image.shape[0] == 3 * 600 and image.shape[1] > 600 and image.dtype == np.uint8
    assert np.all(image[1200:, :600] == 2)  # the third row's rendering is shard 1's
    assert env._visualizer._ender.status(2) == "env 2"
    env.close()


def test_registration_builds_the_device_resident_environment(monkeypatch):
    """make_vec / the gymnasium vector_entry_point build DeviceVectorDiscreteSteps unless glue="host"."""
    from reinfocus_amd import registration
    from reinfocus_amd.environments import harness

    built = []
    monkeypatch.setattr(harness, "DeviceVectorDiscreteSteps", lambda *a, **k: built.append(("device", a, k)) or "dev")
    monkeypatch.setattr(harness, "VectorDiscreteSteps", lambda *a, **k: built.append(("host", a, k)) or "host")
    assert registration.make_vec("DiscreteSteps-v0", 7, vector_kwargs={"render_mode": "rgb_array"}) == "dev"
    assert built[-1] == ("device", (20, 7, "rgb_array"), {})
    assert registration.make_vec("DiscreteSteps-v0", 3, vector_kwargs={"max_episode_steps": 5}, glue="host") == "host"
    assert built[-1] == ("host", (5, 3, None), {})
    assert registration.ENTRY_POINTS["DiscreteSteps-v0"]["vector_entry_point"] == "reinfocus_amd.registration:vector_discrete_steps"
    with pytest.raises(AssertionError):
        registration.make_vec("DiscreteSteps-v0", 2, vectorization_mode="sync")


def test_environments_derive_from_gymnasium_when_it_is_importable(tmp_path):
    """With a gymnasium on the path the environments are gymnasium.Env / VectorEnv subclasses (the
    reference's are: environment.py:19, vector_environment.py:19) and the ids are registered with
    it (examples/__init__.py:6-18).  This image has no gymnasium, so a minimal package with the
    module layout of gymnasium 0.29 stands in; the check runs in a fresh interpreter."""
    pkg = tmp_path / "gymnasium"
    (pkg / "experimental" / "vector").mkdir(parents=True)
    (pkg / "vector").mkdir()
    (pkg / "envs").mkdir()
    (pkg / "__init__.py").write_text("class Env:\n    pass\nfrom gymnasium import spaces\n")
    (pkg / "spaces.py").write_text(textwrap.dedent("""
        import numpy as np
        class Box:
            def __init__(self, low, high, shape=None, dtype=np.float32):
                self.low = np.atleast_1d(np.asarray(low, dtype=dtype)); self.high = np.atleast_1d(np.asarray(high, dtype=dtype))
                self.shape = self.low.shape; self.dtype = np.dtype(dtype)
        class Discrete:
            def __init__(self, n): self.n = n; self.shape = ()
        class MultiDiscrete:
            def __init__(self, nvec): self.nvec = nvec
        """))
    (pkg / "experimental" / "__init__.py").write_text("")
    (pkg / "experimental" / "vector" / "__init__.py").write_text("class VectorEnv:\n    pass\n")
    (pkg / "vector" / "__init__.py").write_text("")
    (pkg / "vector" / "utils.py").write_text("def batch_space(space, n=1):\n    return ('batched', space, n)\n")
    (pkg / "envs" / "__init__.py").write_text("")
    (pkg / "envs" / "registration.py").write_text(
        "registry = {}\ndef register(id, **spec):\n    registry[id] = spec\n")
    code = textwrap.dedent("""
        import gymnasium
        from gymnasium.experimental.vector import VectorEnv
        from gymnasium.envs import registration as gym_registration
        from reinfocus_amd import registration
        from reinfocus_amd.environments import harness, spaces
        assert spaces.HAVE_GYMNASIUM
        for cls in (harness.VectorDiscreteSteps, harness.DeviceVectorDiscreteSteps, harness.ShardedVectorDiscreteSteps):
            assert issubclass(cls, VectorEnv), cls
        for cls in (harness.DiscreteSteps, harness.ContinuousJumps):
            assert issubclass(cls, gymnasium.Env) and not issubclass(cls, VectorEnv), cls
        spec = gym_registration.registry["DiscreteSteps-v0"]
        assert spec["max_episode_steps"] == 20 and spec["vector_entry_point"].endswith(":vector_discrete_steps")
        assert spec["entry_point"].endswith(":DiscreteSteps")
        assert gym_registration.registry["ContinuousJumps-v0"]["max_episode_steps"] == 20
        print("ok")
        """)
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([str(tmp_path), ROOT]))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0 and out.stdout.strip() == "ok", out.stderr[-3000:]
