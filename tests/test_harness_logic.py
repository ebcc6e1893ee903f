"""Host logic of the DiscreteSteps-v0 harness with the GPU pieces mocked out, in the
style of the reference's tests/environments/vector_environment_test.py (strategy objects
replaced by mocks) -- call order, auto-reset contract, transformer / ender / rewarder /
normaliser arithmetic.  The hot path itself is covered by the -m gpu tests."""

import numpy as np
import pytest


class FakeRenderer:
    """Stands in for FastRenderer: records the calls FocusObserver makes."""

    def __init__(self):
        self._samples_per_pixel = 100
        self._ctx = type("Ctx", (), {"device": 0, "close": lambda self: None})()
        self.calls = []
        self.targets = None
        self.focus = None

    def update_targets(self, targets):
        self.calls.append(("targets", len(targets)))
        self.targets = np.asarray(targets, dtype=np.float32)

    def update_focus_planes(self, focus):
        self.calls.append(("focus", len(focus)))
        self.focus = np.asarray(focus, dtype=np.float32)

    def render(self, height):
        self.calls.append(("render", height))
        return ("frames", self.targets.copy(), self.focus.copy())


@pytest.fixture()
def env(monkeypatch):
    from reinfocus_amd import vision
    from reinfocus_amd.environments import harness, state_observer
    from reinfocus_amd.graphics import render

    fake = FakeRenderer()
    monkeypatch.setattr(render, "FastRenderer", lambda **kw: fake)
    monkeypatch.setattr(state_observer, "cached_focus_extrema", lambda *a, **k: (30.0, 350.0))
    # focus value falls linearly with |target - focus|
    monkeypatch.setattr(vision, "focus_values", lambda fr: list(350.0 - 60.0 * np.abs(fr[1] - fr[2]).astype(np.float64)))
    e = harness.VectorDiscreteSteps(num_envs=6, seed=5)
    e._fake = fake
    return e


def test_spaces(env):
    assert env.single_action_space.n == 13
    assert env.observation_space.shape == (6, 4)
    assert np.all(env.single_observation_space.low == -1) and np.all(env.single_observation_space.high == 1)
    assert env.render_mode is None and env.render() is None


def test_reset_call_order_and_observation(env):
    obs, info = env.reset()
    assert info == {} and obs.shape == (6, 4) and obs.dtype == np.float32
    assert env._fake.calls == [("targets", 6), ("focus", 6), ("render", 300)]
    state = env._state
    assert state.dtype == np.float32 and np.all((state >= 5) & (state <= 10))
    # [focus plane, focus value] normalised, zero deltas (state_observer.py:263-292, :508-517)
    assert np.allclose(obs[:, 0], (state[:, 1] - 7.5) / 2.5, atol=1e-6)
    fv = 350.0 - 60.0 * np.abs(state[:, 0] - state[:, 1])
    assert np.allclose(obs[:, 1], np.clip((fv - 190.0) / 160.0, -1, 1), atol=1e-5)
    assert np.all(obs[:, 2:] == 0)


def test_step_arithmetic(env):
    env.reset(state=[[6.0, 9.0], [6.0, 6.1], [9.0, 5.0], [7.0, 7.0], [5.0, 10.0], [8.0, 8.5]])
    env._fake.calls.clear()
    actions = np.array([0, 6, 12, 8, 5, 7])  # -5, 0, +5, +0.3125, -0.15625, +0.15625
    obs, rewards, terminated, truncated, info = env.step(actions)
    want_focus = np.array([5.0, 6.1, 10.0, 7.3125, 9.84375, 8.65625], dtype=np.float32)
    assert np.array_equal(env._state[:, 1], want_focus)                 # clipped to [5, 10]
    assert np.array_equal(env._state[:, 0], [6, 6, 9, 7, 5, 8])         # targets untouched
    assert env._fake.calls == [("targets", 6), ("focus", 6), ("render", 300)]
    assert not terminated.any() and not truncated.any()
    # observation: [pos, fv, dpos / 5, dfv / (max - min)] normalised
    assert np.allclose(obs[:, 2], (want_focus - [9, 6.1, 5, 7, 10, 8.5]) / 5.0, atol=1e-6)
    # reward = -|move| / 0.5 + normalised fv + [|target - focus| < 0.25]
    moved = np.abs(want_focus - np.array([9, 6.1, 5, 7, 10, 8.5], dtype=np.float32)) * -1.0 / 0.5
    on_target = (np.abs(env._state[:, 0] - want_focus) < 0.25).astype(np.float64)
    assert np.allclose(rewards, moved + obs[:, 1] + on_target, atol=1e-6)
    assert on_target.tolist() == [0, 1, 0, 0, 0, 0]


def test_diverging_ender_and_auto_reset(env):
    env.reset(state=[[7.5, 7.5]] * 6)
    # env 0 diverges three times (|diff| grows by > 0.125 each step); the others stay put
    seq = [9, 9, 9]  # +0.625 each step
    for k, a in enumerate(seq):
        env._fake.calls.clear()
        actions = np.full(6, 6)
        actions[0] = a
        obs, rewards, terminated, truncated, _ = env.step(actions)
        if k < 2:
            assert not truncated.any()
            assert env._fake.calls == [("targets", 6), ("focus", 6), ("render", 300)]
    assert truncated.tolist() == [True] + [False] * 5
    # same-step auto reset: a second, partial render of just the done env (vector_environment.py:137-151)
    assert env._fake.calls == [("targets", 6), ("focus", 6), ("render", 300),
                               ("targets", 1), ("focus", 1), ("render", 300)]
    assert np.all(obs[0, 2:] == 0)                       # fresh episode: zero deltas
    assert env._state[0, 1] != np.float32(7.5 + 3 * 0.625)  # state was re-initialised
    assert env._ender._diverging_steps[0] == 0 and env._ender._steps[0] == 0
    assert np.all(env._ender._steps[1:] == 3)


def test_time_limit(env):
    env.reset(state=[[7.5, 7.5]] * 6)
    for step in range(20):
        _, _, terminated, truncated, _ = env.step(np.full(6, 6))
        assert truncated.all() == (step == 19)
    assert not terminated.any()
    assert np.all(env._ender._steps == 0)


def test_registration_local_make_vec(monkeypatch, env):
    from reinfocus_amd import registration

    # examples/__init__.py:6-18
    assert set(registration.ENTRY_POINTS) == {"DiscreteSteps-v0", "ContinuousJumps-v0"}
    assert registration.ENTRY_POINTS["DiscreteSteps-v0"]["max_episode_steps"] == 20
    assert "vector_entry_point" not in registration.ENTRY_POINTS["ContinuousJumps-v0"]
    with pytest.raises(KeyError):
        registration.make_vec("ContinuousJumps-v0", 2)
    with pytest.raises(KeyError):
        registration.make("NoSuchEnv-v0")


def test_continuous_jumps_arithmetic(env):
    """ContinuousJumpTransformer + ObservationRewarder + StoppedRewarder * OnTargetRewarder."""
    from reinfocus_amd import registration

    cj = registration.make("ContinuousJumps-v0")
    assert cj.action_space.shape == (1,) and cj.action_space.low[0] == -1 and cj.action_space.high[0] == 1
    obs, _ = cj.reset(state=[[7.0, 9.0]])
    assert obs.shape == (4,) and obs.dtype == np.float32
    # action 0 -> position 7.5: a real jump, not stopped, not on target (|7 - 7.5| = 0.5)
    obs, reward, terminated, truncated, _ = cj.step(np.float32(0.0))
    assert np.array_equal(cj._state, np.array([[7.0, 7.5]], dtype=np.float32))
    assert reward == np.float64(obs[1]) and not terminated and not truncated
    # action -0.38 -> 6.55?  |7.5 - 6.55| > 0.125: jump; on target? |7 - 6.55| = 0.45: no
    obs, reward, *_ = cj.step(np.float32(-0.38))
    assert abs(cj._state[0, 1] - 6.55) < 1e-6 and reward == np.float64(obs[1])
    # a jump of less than 0.125 is suppressed: stopped, and now |7 - 6.55| >= 0.25 -> reward = fv
    obs, reward, *_ = cj.step(np.float32(-0.36))
    assert abs(cj._state[0, 1] - 6.55) < 1e-6 and reward == np.float64(obs[1])
    # jump onto the target, then stay: stopped * on_target adds 1
    cj.step(np.float32(-0.2))
    assert abs(cj._state[0, 1] - 7.0) < 1e-6
    obs, reward, *_ = cj.step(np.float32(-0.21))
    assert abs(cj._state[0, 1] - 7.0) < 1e-6 and reward == np.float64(obs[1]) + 1.0


@pytest.mark.parametrize("seed", [0, 1, 12345])
@pytest.mark.parametrize("n", [1, 8, 257])
def test_proposed_rows_are_the_rows_the_initializer_draws(seed, n):
    """_Initializer.propose (the device-resident step's candidate rows) returns what initialize would, bit for bit, and
    consumes nothing: state_initializer.py:30-71 draws with Generator.uniform, vector_environment.py:144 only for the
    environments that ended."""
    from reinfocus_amd.environments import harness

    ends = harness._DeviceShard.ENDS
    a, b = harness._Initializer(ends, seed), harness._Initializer(ends, seed)
    for _ in range(5):
        rows = a.propose(n)
        assert rows.dtype == np.float32 and rows.shape == (n, 2)
        assert np.array_equal(rows, a.propose(n))  # nothing consumed
        used = int(np.random.default_rng(seed + n).integers(0, n + 1))
        want = b.initialize(n if used == 0 else used)
        if used:
            assert np.array_equal(a.initialize(used), want)
            assert np.array_equal(rows[:used], want)
        else:
            assert np.array_equal(rows, want)
            a.initialize(n)
