"""stable-baselines3 adapter, protocol level: the reference has
no tests for experimental/vector_shim.py, so these pin the behaviour its code spells out
(vector_shim.py:48-184) on a scripted vector environment; tests/test_gpu_environment.py runs
the adapter over the real device-resident environment."""

import numpy as np
import pytest

from reinfocus_amd.environments import harness, spaces, vector_shim


class _Scripted(harness.VectorDiscreteSteps):
    """A VectorDiscreteSteps that never touches the GPU: canned step results."""

    def __init__(self, num_envs):  # pylint: disable=super-init-not-called
        self.num_envs = num_envs
        self.single_observation_space = spaces.Box(-np.ones(4, dtype=np.float32), np.ones(4, dtype=np.float32),
                                                   dtype=np.float32)
        self.single_action_space = spaces.Discrete(13)
        self.render_mode = "rgb_array"
        self.closed = False
        self.seen = []

    def reset(self, *, seed=None, options=None, state=None):
        return np.zeros((self.num_envs, 4), dtype=np.float32), {"ignored": 1}

    def step(self, actions):
        self.seen.append(np.asarray(actions).copy())
        obs = np.arange(self.num_envs * 4, dtype=np.float32).reshape(self.num_envs, 4)
        rewards = np.linspace(-1, 1, self.num_envs)
        terminated = np.array([False, False, True][:self.num_envs])
        truncated = np.array([False, True, False][:self.num_envs])
        info = {"steps": np.arange(self.num_envs), "label": "not an array"}
        return obs, rewards, terminated, truncated, info

    def render(self):
        return np.full((2, 2, 3), 9, dtype=np.uint8)

    def close(self):
        self.closed = True


def test_sb3_wrapper_protocol():
    env = _Scripted(3)
    testee = vector_shim.SB3Wrapper(env, "rgb_array")
    assert testee.num_envs == 3 and testee.render_mode == "rgb_array"
    assert testee.observation_space is env.single_observation_space
    assert testee.action_space is env.single_action_space

    obs = testee.reset()  # stable-baselines3: observations only
    assert isinstance(obs, np.ndarray) and obs.shape == (3, 4)

    with pytest.raises(AssertionError):
        testee.step_wait()  # no step_async yet
    actions = np.array([1, 2, 3])
    testee.step_async(actions)
    assert env.seen == []  # nothing happens until step_wait
    obs, rewards, dones, infos = testee.step_wait()
    assert np.array_equal(env.seen[0], actions)
    assert list(dones) == [False, True, True]  # terminated | truncated
    assert [sorted(i) for i in infos] == [["steps"], ["steps", "terminal_observation"],
                                          ["steps", "terminal_observation"]]
    assert [i["steps"] for i in infos] == [0, 1, 2]
    assert np.array_equal(infos[2]["terminal_observation"], obs[2])
    assert len(rewards) == 3

    assert testee.get_attr("num_envs") == [3, 3, 3]
    assert testee.get_attr("num_envs", 0) == [3]
    assert testee.get_attr("num_envs", [0, 2]) == [3, 3]
    with pytest.raises(NotImplementedError):
        testee.get_attr("no_such_attribute")
    with pytest.raises(NotImplementedError):
        testee.set_attr("num_envs", 4)
    with pytest.raises(NotImplementedError):
        testee.env_method("reset")
    assert testee.env_is_wrapped(object) == [False] * 3
    assert testee.env_is_wrapped(object, indices=1) == [False]
    images = testee.get_images()
    assert len(images) == 1 and images[0].shape == (2, 2, 3)
    testee.close()
    assert env.closed


def test_sb3_wrapper_rejects_other_environments():
    with pytest.raises(NotImplementedError):
        vector_shim.SB3Wrapper(object(), None)


def test_rewrapper_needs_stable_baselines3():
    try:
        import stable_baselines3  # noqa: F401
    except ImportError:
        with pytest.raises(ImportError, match="stable-baselines3"):
            vector_shim.rewrapper(object())
    else:  # pragma: no cover - not installed in this image
        assert vector_shim.rewrapper("not a DummyVecEnv") == "not a DummyVecEnv"
