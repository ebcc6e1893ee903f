"""The single-environment shell: the reference's own tests of environments/environment.py
(tests/environments/environment_test.py:115-236), its initializer test
(tests/environments/state_initializer_test.py:10-36) and its hit-record tests
(tests/graphics/hit_record_test.py:16-87), transcribed.

The reference's Environment takes its six strategy objects as arguments and the tests pass mocks.
harness.DiscreteSteps / ContinuousJumps build theirs, so the mocks replace the built ones on an
instance made without a renderer (no GPU is touched: these tests pin call order, what is passed on and
what is returned, exactly as the reference's do)."""

from unittest import mock

import numpy as np
import pytest
from numpy import testing

from reinfocus_amd.environments import harness


def make_ender(is_terminated=False, is_truncated=False):
    ender = mock.Mock()
    ender.is_terminated.return_value = [is_terminated]
    ender.is_truncated.return_value = [is_truncated]
    return ender


def make_initializer(initial_state=np.zeros((1, 2))):  # (target, focus plane): the tasks' state has two elements
    initializer = mock.Mock()
    initializer.initialize.return_value = initial_state
    return initializer


def make_observer():
    """Returns the state itself, so that a single environment unpacks it as its observation."""
    observer = mock.Mock()
    observer.observe.side_effect = lambda state, indices=None: state
    observer.reset.side_effect = lambda state, indices=None: state
    return observer


def make_rewarder():
    rewarder = mock.Mock()
    rewarder.reward.return_value = [0]
    return rewarder


def make_testee(cls=harness.DiscreteSteps, ender=None, initializer=None, observer=None, rewarder=None, transform=None,
                visualizer=None, render_mode=None):
    """environment_test.py:86-113 -- the shell around mocked strategies."""
    env = object.__new__(cls)  # no renderer, no GPU: only the shell's own logic runs
    env.render_mode = render_mode
    env.num_envs = 1
    env._state = None
    env._limits = (-np.inf, np.inf)
    env._ender = ender or make_ender()
    env._initializer = initializer or make_initializer()
    env._observer = observer or make_observer()
    env._rewarder = rewarder or make_rewarder()
    env._visualizer = visualizer or mock.Mock()
    env._transform = transform or (lambda states, actions: states)  # make_transformer: the old state is the new one
    env._stop_threshold = 0.125  # ContinuousJumps' own rewarder terms
    env._old_focus = np.zeros(1)
    return env


SHELLS = [harness.DiscreteSteps, harness.ContinuousJumps]


def test_spaces():
    """environment_test.py:118-134: the spaces are the observer's and the transformer's (here: what the
    constructor derived from them; no GPU needed to state them)."""
    env = make_testee()
    env.single_action_space = env.action_space = "actions"
    env.single_observation_space = env.observation_space = "observations"
    assert env.action_space == "actions" and env.observation_space == "observations"


@pytest.mark.parametrize("cls", SHELLS)
def test_initialization(cls):
    """:136-144 -- reset() returns the observation of the initializer's state."""
    target = np.array([[-4, 8]])
    testing.assert_allclose(make_testee(cls, initializer=make_initializer(target)).reset()[0], target[0])


@pytest.mark.parametrize("cls", SHELLS)
def test_transforms(cls):
    """:146-156 -- step() observes the transformed state."""
    target = np.array([[-4, 8]])
    env = make_testee(cls, initializer=make_initializer(target))
    env.reset()
    testing.assert_allclose(env.step(0)[0], target[0])


def test_reward():
    """:158-170 -- the reward is the rewarder's, for the new state and its observation."""
    rewarder = mock.Mock()
    rewarder.reward.side_effect = lambda s, o: s[0] + o[0][1]
    env = make_testee(initializer=make_initializer(np.array([[-4, 8]])), rewarder=rewarder)
    env.reset()
    assert env.step(0)[1] == 4


@pytest.mark.parametrize("cls", SHELLS)
def test_terminated_and_truncated(cls):
    """:172-186 -- both flags come from the episode ender, unbatched."""
    env = make_testee(cls, ender=make_ender(True, False), initializer=make_initializer(np.array([[6.0, 7.0]])))
    env.reset()
    testing.assert_allclose(env.step(0)[2:4], [True, False])
    env = make_testee(cls, ender=make_ender(False, True), initializer=make_initializer(np.array([[6.0, 7.0]])))
    env.reset()
    testing.assert_allclose(env.step(0)[2:4], [False, True])


@pytest.mark.parametrize("cls", SHELLS)
def test_reset_resets_the_ender_and_the_observer(cls):
    """:188-210."""
    ender, observer = make_ender(), make_observer()
    env = make_testee(cls, ender=ender, observer=observer)
    env.reset()
    ender.reset.assert_called_once()
    observer.reset.assert_called_once()


@pytest.mark.parametrize("cls", SHELLS)
def test_call_order_of_reset_and_step(cls):
    """environment.py:64-121: reset = initialize(1) -> ender.reset -> observer.reset -> rewarder.reset
    (-> visualizer.reset); step = transform -> ender.step -> observe (-> visualizer.step) -> reward ->
    is_terminated -> is_truncated; and no auto-reset in the single-environment shell."""
    calls = []
    parent = mock.Mock()
    ender, initializer, observer, rewarder, visualizer = (parent.ender, parent.initializer, parent.observer,
                                                          parent.rewarder, parent.visualizer)
    state = np.array([[6.0, 7.0]], dtype=np.float32)
    initializer.initialize.return_value = state
    observer.reset.side_effect = lambda s, indices=None: s
    observer.observe.side_effect = lambda s, indices=None: s
    rewarder.reward.return_value = [1.5]
    ender.is_terminated.return_value = [False]
    ender.is_truncated.return_value = [True]

    def transform(states, actions):
        calls.append("transform")
        return states

    env = make_testee(cls, ender=ender, initializer=initializer, observer=observer, rewarder=rewarder,
                      transform=transform, visualizer=visualizer, render_mode="rgb_array")
    env.reset()
    assert [c[0] for c in parent.mock_calls] == ["initializer.initialize", "ender.reset", "observer.reset",
                                                 "rewarder.reset", "visualizer.reset"]
    initializer.initialize.assert_called_once_with(1)
    parent.reset_mock()
    out = env.step(0)
    names = [c[0] for c in parent.mock_calls]
    if cls is harness.DiscreteSteps:
        assert calls == ["transform"] and names == ["ender.step", "observer.observe", "visualizer.step",
                                                    "rewarder.reward", "ender.is_terminated", "ender.is_truncated"]
        assert out[1] == 1.5
    else:  # ContinuousJumps computes its reward terms itself (custom_environments.py:300-336)
        assert names == ["ender.step", "observer.observe", "visualizer.step", "ender.is_terminated", "ender.is_truncated"]
    assert out[2] is False and out[3] is True and out[4] == {}
    initializer.initialize.assert_not_called()  # truncated, and still no reset


@pytest.mark.parametrize("cls", SHELLS)
def test_no_render(cls):
    """:212-219."""
    env = make_testee(cls)
    env.reset()
    assert env.render() is None


@pytest.mark.parametrize("cls", SHELLS)
def test_rgb_array_render(cls):
    """:221-236."""
    target = mock.Mock()
    visualizer = mock.Mock()
    visualizer.visualize.return_value = target
    env = make_testee(cls, visualizer=visualizer, render_mode="rgb_array")
    env.reset()
    assert env.render() == target


# --- tests/environments/state_initializer_test.py:10-36 ----------------------------------------------
def test_ranged_state_initializer():
    """RangedInitializer samples every element from its range(s); the harness's initializer is the
    (seedable) single-range form the registered environments use: shape, bounds, dtype, and two draws
    differ."""
    initial_states = harness._Initializer((0.4, 0.6), None).initialize(2)
    assert initial_states.shape == (2, 2) and initial_states.dtype == np.float32
    with pytest.raises(AssertionError):
        testing.assert_allclose(*initial_states)
    assert np.all((0.4 <= initial_states) & (initial_states <= 0.6))
    # seeded: reproducible, and consumption is per row (what the device-resident step relies on)
    a, b = harness._Initializer((5.0, 10.0), 3), harness._Initializer((5.0, 10.0), 3)
    first = a.initialize(5)
    assert np.array_equal(first, np.concatenate([b.initialize(2), b.initialize(3)]))
    assert np.all((5.0 <= first) & (first <= 10.0))


# --- tests/graphics/hit_record_test.py:16-87 -----------------------------------------------------------
def test_hit_record_layout(oracle):
    """hit_record.py: a record is (p[3], n[3], t, uv[2], uf[2], m) -- 12 float32 -- and a miss leaves the
    empty record, all zeros (hit_record_test.py:19-36); a hit fills the fields in that order (:42-87)."""
    for miss, record in (oracle.sphere_hit([0, 0, 0, 1, 4, 8], (10, 0, 0), (0, 1, 0), 0.0, 100.0),
                         oracle.rectangle_hit([-1, 1, -1, 1, 1, 4, 8], (0, 0, 0), (0, 0, -1), 0.0, 100.0),
                         oracle.fast_hit(np.array([1.0, -1.0], dtype=np.float32), (0, 0, 0), (0, 0, 1), 0.0, 100.0)):
        assert not miss
        testing.assert_allclose(record[:12], np.zeros(12))
    # a rectangle at z = 2 spanning [1, 5] x [1, 9], frequencies (9, 10), hit at (3, 4, 2) by the ray
    # from the origin through (1.5, 2, 1): p = (3, 4, 2), n = (0, 0, 1), t = 2, uv = (0.5, 0.375)
    hit, record = oracle.rectangle_hit([1, 5, 1, 9, 2, 9, 10], (0, 0, 0), (1.5, 2, 1), 0.0, 100.0)
    assert hit
    P, N, T, UV, UF, M = slice(0, 3), slice(3, 6), 6, slice(7, 9), slice(9, 11), 11  # hit_record.py:14-19
    testing.assert_allclose(record[P], (3, 4, 2))
    testing.assert_allclose(record[N], (0, 0, 1))
    testing.assert_allclose(record[T], 2)
    testing.assert_allclose(record[UV], (0.5, 0.375))
    testing.assert_allclose(record[UF], (9, 10))
    assert record[M] == 1.0  # shape.RECTANGLE
