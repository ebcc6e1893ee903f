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


def test_rewrapper_with_stable_baselines3_on_the_path(tmp_path):
    """vector_shim.rewrapper (reference: experimental/vector_shim.py:187-229) needs stable-baselines3,
    which this image does not have: a minimal package with SB3's module layout and the classes the
    function touches (DummyVecEnv, VecEnv, Monitor, VecMonitor) stands in, in a fresh interpreter.
    Checked: anything but a DummyVecEnv of a registered environment passes through unchanged; a
    DummyVecEnv of n Monitor-wrapped DiscreteSteps-v0 members becomes VecMonitor(SB3Wrapper(ONE
    vector environment of n members)) built through the env id's vector entry point with the
    spec's max_episode_steps and render_mode "human" -> "rgb_array"; SB3Wrapper derives from SB3's
    VecEnv."""
    import os
    import subprocess
    import sys
    import textwrap

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pkg = tmp_path / "stable_baselines3"
    (pkg / "common" / "vec_env").mkdir(parents=True)
    (pkg / "__init__.py").write_text("")
    (pkg / "common" / "__init__.py").write_text("")
    (pkg / "common" / "monitor.py").write_text(textwrap.dedent("""
        class Monitor:
            EXT = "monitor.csv"
            def __init__(self, env, info_keywords=()):
                self.env, self.info_keywords = env, info_keywords
                self.spec, self.render_mode = env.spec, env.render_mode
        """))
    (pkg / "common" / "vec_env" / "base_vec_env.py").write_text(textwrap.dedent("""
        class VecEnv:
            def __init__(self, num_envs, observation_space, action_space):
                self.num_envs, self.observation_space, self.action_space = num_envs, observation_space, action_space
                self.render_mode = None
            def step(self, actions):
                self.step_async(actions)
                return self.step_wait()
        """))
    (pkg / "common" / "vec_env" / "vec_monitor.py").write_text(textwrap.dedent("""
        class VecMonitor:
            def __init__(self, venv, filename=None, info_keywords=()):
                self.venv, self.filename, self.info_keywords = venv, filename, info_keywords
        """))
    (pkg / "common" / "vec_env" / "__init__.py").write_text(textwrap.dedent("""
        from stable_baselines3.common.vec_env.base_vec_env import VecEnv
        class DummyVecEnv(VecEnv):
            def __init__(self, envs):
                self.envs = envs
                self.num_envs = len(envs)
        """))
    code = textwrap.dedent("""
        import types
        import numpy as np
        from stable_baselines3.common import monitor, vec_env
        from stable_baselines3.common.vec_env import base_vec_env, vec_monitor
        from reinfocus_amd import registration
        from reinfocus_amd.environments import vector_shim
        from tests.test_vector_shim import _Scripted

        assert issubclass(vector_shim.SB3Wrapper, base_vec_env.VecEnv)
        asked = []
        def make_vec(env_id, num_envs, **kwargs):
            asked.append((env_id, num_envs, kwargs))
            return _Scripted(num_envs)
        registration.make_vec = make_vec

        not_dummy = object()
        assert vector_shim.rewrapper(not_dummy) is not_dummy
        def member(spec, render_mode="human"):
            return types.SimpleNamespace(spec=spec, render_mode=render_mode)
        no_spec = vec_env.DummyVecEnv([member(None)])
        assert vector_shim.rewrapper(no_spec) is no_spec and asked == []

        spec = types.SimpleNamespace(id="DiscreteSteps-v0", max_episode_steps=7)
        members = [monitor.Monitor(member(spec), info_keywords=("k",)) for _ in range(3)]
        result = vector_shim.rewrapper(vec_env.DummyVecEnv(members))
        assert asked == [("DiscreteSteps-v0", 3, {"max_episode_steps": 7, "render_mode": "rgb_array"})]
        assert isinstance(result, vec_monitor.VecMonitor)
        assert result.filename == "monitor.csv" and result.info_keywords == ("k",)
        wrapper = result.venv
        assert isinstance(wrapper, vector_shim.SB3Wrapper) and wrapper.num_envs == 3 and wrapper.render_mode == "rgb_array"
        observations, rewards, dones, infos = wrapper.step(np.array([1, 2, 3]))
        assert observations.shape == (3, 4) and list(dones) == [False, True, True]

        plain = vector_shim.rewrapper(vec_env.DummyVecEnv([member(types.SimpleNamespace(id="DiscreteSteps-v0",
                                                                                      max_episode_steps=None), None)]))
        assert isinstance(plain, vector_shim.SB3Wrapper) and plain.render_mode is None
        assert asked[-1] == ("DiscreteSteps-v0", 1, {"render_mode": None})
        print("ok")
        """)
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([str(tmp_path), root]), REINFOCUS_NO_AUTOBUILD="1")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=300, cwd=root)
    assert out.returncode == 0 and out.stdout.strip() == "ok", out.stderr[-3000:]
