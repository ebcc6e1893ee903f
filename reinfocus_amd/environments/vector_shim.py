"""stable-baselines3 adapter for the vector environments.

Mirror of reinfocus/environments/experimental/vector_shim.py: `SB3Wrapper` (:20-184) turns
the gymnasium-style vector API (reset -> (obs, info); step -> (obs, rewards, terminated,
truncated, info)) into stable-baselines3's VecEnv protocol (reset -> obs; step_async /
step_wait -> (obs, rewards, dones, infos)), and `rewrapper` (:187-229) swaps the DummyVecEnv
that rl_zoo3 builds for one vector environment with `num_envs` members.

stable-baselines3 and gymnasium are not installed in this image.  When stable-baselines3 is
importable SB3Wrapper derives from its VecEnv (so VecMonitor / VecNormalize accept it);
otherwise it derives from a minimal stand-in with the same protocol so that the adapter
logic is testable here.  The wrapped environment may be `harness.VectorDiscreteSteps` or the
device-resident `harness.DeviceVectorDiscreteSteps` (one rf_env_step per step_wait).
"""

from collections.abc import Iterable

import numpy as np

from reinfocus_amd.environments import harness

try:  # pragma: no cover - not installed in this image
    from stable_baselines3.common.vec_env import base_vec_env as _sb3_base

    _VecEnvBase = _sb3_base.VecEnv
except ImportError:

    class _VecEnvBase:
        """The part of stable_baselines3.common.vec_env.base_vec_env.VecEnv the shim uses."""

        def __init__(self, num_envs, observation_space, action_space):
            self.num_envs = num_envs
            self.observation_space = observation_space
            self.action_space = action_space
            self.render_mode = None

        def step(self, actions):
            self.step_async(actions)
            return self.step_wait()


_VECTOR_ENVIRONMENTS = (harness.VectorDiscreteSteps, harness.DeviceVectorDiscreteSteps)


def _how_many(indices, num_envs):
    """Number of member environments `indices` (None, an int or an iterable of ints) selects."""
    if indices is None:
        return num_envs
    if isinstance(indices, int):
        return 1
    if isinstance(indices, Iterable):
        return sum(1 for _ in indices)
    return num_envs


class SB3Wrapper(_VecEnvBase):
    """stable-baselines3 VecEnv over one vector environment of this package
    (vector_shim.py:20-184)."""

    def __init__(self, env, render_mode):
        if not isinstance(env, _VECTOR_ENVIRONMENTS):  # vector_shim.py:32-33
            raise NotImplementedError
        super().__init__(env.num_envs, env.single_observation_space, env.single_action_space)
        self._env = env
        self._pending_actions = None
        self.render_mode = render_mode

    # -- stepping -------------------------------------------------------------------------------
    def reset(self):
        observations, _info = self._env.reset()
        return observations

    def step_async(self, actions):
        self._pending_actions = actions

    def step_wait(self):
        """vector_shim.py:63-93.  The environments reset themselves inside step(), so -- as in the
        reference -- "terminal_observation" is the observation step() returned for that
        environment, i.e. already the first one of its next episode."""
        assert self._pending_actions is not None
        observations, rewards, terminated, truncated, info = self._env.step(self._pending_actions)
        dones = terminated | truncated
        per_env_keys = [key for key, value in info.items() if isinstance(value, np.ndarray)]
        infos = [{key: info[key][i] for key in per_env_keys} for i in range(self.num_envs)]
        for i in np.flatnonzero(dones):
            infos[i]["terminal_observation"] = observations[i]
        return observations, rewards, dones, infos

    def close(self):
        self._env.close()

    # -- attribute / method plumbing stable-baselines3 expects ------------------------------------
    def get_attr(self, attr_name, indices=None):
        if not hasattr(self._env, attr_name):
            raise NotImplementedError(f"{attr_name}, {indices}")
        return [getattr(self._env, attr_name)] * _how_many(indices, self._env.num_envs)

    def set_attr(self, attr_name, value, indices=None):
        raise NotImplementedError(f"{attr_name}, {value}, {indices}")

    def env_method(self, method_name, *method_args, indices=None, **method_kwargs):
        raise NotImplementedError(f"{method_name}, {method_args}, {indices}, {method_kwargs}")

    def env_is_wrapped(self, wrapper_class, indices=None):
        return [False] * _how_many(indices, self._env.num_envs)

    def get_images(self):
        return [self._env.render()]


def rewrapper(naive_vec_env):
    """vector_shim.py:187-229: replaces the DummyVecEnv stable-baselines3 built (one Python
    environment per member) by ONE vector environment of the same id with as many members.
    Needs stable-baselines3 and gymnasium; anything that is not a DummyVecEnv of a registered
    environment is returned unchanged."""
    try:
        from stable_baselines3.common import monitor, vec_env
        from stable_baselines3.common.vec_env import vec_monitor
    except ImportError as error:  # no silent fallback: the caller asked for an SB3 object
        raise ImportError("rewrapper needs stable-baselines3 (not installed in this image)") from error
    from reinfocus_amd import registration

    if not isinstance(naive_vec_env, vec_env.DummyVecEnv):
        return naive_vec_env
    wrapper = naive_vec_env.envs[0]
    if wrapper.spec is None:
        return naive_vec_env
    vector_kwargs = {}
    if wrapper.spec.max_episode_steps is not None:
        vector_kwargs["max_episode_steps"] = wrapper.spec.max_episode_steps
    render_mode = "rgb_array" if wrapper.render_mode == "human" else None
    vector_kwargs["render_mode"] = render_mode
    result = SB3Wrapper(registration.make_vec(wrapper.spec.id, naive_vec_env.num_envs, **vector_kwargs), render_mode)
    if isinstance(wrapper, monitor.Monitor):
        result = vec_monitor.VecMonitor(result, wrapper.EXT, wrapper.info_keywords)
    return result
