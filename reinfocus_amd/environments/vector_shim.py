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


class SB3Wrapper(_VecEnvBase):
    def __init__(self, env, render_mode):
        """vector_shim.py:25-46: only this package's vector environments are accepted."""
        if not isinstance(env, _VECTOR_ENVIRONMENTS):
            raise NotImplementedError
        self._env = env
        super().__init__(env.num_envs, env.single_observation_space, env.single_action_space)
        self.render_mode = render_mode
        self._actions = None

    def step_async(self, actions):
        self._actions = actions

    def reset(self):
        return self._env.reset()[0]

    def step_wait(self):
        """vector_shim.py:63-93.  The environments reset themselves inside step(), so -- as in
        the reference -- "terminal_observation" is the observation step() returned for that
        environment, i.e. already the first one of its next episode."""
        assert self._actions is not None
        obs, rewards, terminated, truncated, info_dict = self._env.step(self._actions)
        dones = terminated | truncated
        infos = []
        for i in range(self.num_envs):
            infos.append({key: value[i] for key, value in info_dict.items() if isinstance(value, np.ndarray)})
            if dones[i]:
                infos[i]["terminal_observation"] = obs[i]
        return obs, rewards, dones, infos

    def close(self):
        self._env.close()

    def get_attr(self, attr_name, indices=None):
        if hasattr(self._env, attr_name):
            return [getattr(self._env, attr_name)] * self._get_result_length(indices)
        raise NotImplementedError(f"{attr_name}, {indices}")

    def set_attr(self, attr_name, value, indices=None):
        raise NotImplementedError(f"{attr_name}, {value}, {indices}")

    def env_method(self, method_name, *method_args, indices=None, **method_kwargs):
        raise NotImplementedError(f"{method_name}, {method_args}, {indices}, {method_kwargs}")

    def env_is_wrapped(self, wrapper_class, indices=None):
        return [False] * self._get_result_length(indices)

    def _get_result_length(self, indices):
        if isinstance(indices, int):
            return 1
        if isinstance(indices, Iterable):
            return len(list(indices))
        return self._env.num_envs

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
