"""Thin vector-environment harness for the DiscreteSteps-v0 task.

The reference assembles this environment from six strategy objects
(examples/custom_environments.py:114-241 on top of
reinfocus/environments/vector_environment.py:19-176).  Only one of them touches the
GPU (FocusObserver); the others are O(N) numpy glue that SURVEY.md section 8 marks out
of scope for re-implementation.  The benchmark still has to *step* an environment, so
this module reproduces the reference's call order and numpy arithmetic for exactly that
task, vectorised, in one place:

    step:  transform -> ender.step -> observe (render + focus on the GPU) -> reward ->
           terminated/truncated -> same-step auto-reset of the done envs (partial render)

State is float32[N, 2] = [target position, focus plane]; observations are
float32[N, 4] = [focus plane, focus value, their changes], normalised to [-1, 1].
Deliberate difference: the reference's RangedInitializer is unseeded
(state_initializer.py:50); here `seed` makes runs reproducible.
"""

import numpy as np

from reinfocus_amd.environments import episode_visualizer
from reinfocus_amd.environments import spaces
from reinfocus_amd.environments import state_observer
from reinfocus_amd.graphics import render

TARGET, FOCUS = 0, 1  # state element indices (custom_environments.py:169-171)


def _gymnasium_bases():
    """(Env base, VectorEnv base): gymnasium's classes when gymnasium is importable -- the
    reference's environments derive from gymnasium.Env (environments/environment.py:19) and
    gymnasium.experimental.vector.VectorEnv (environments/vector_environment.py:19; plain
    gymnasium.vector.VectorEnv from gymnasium 1.0 on), which gymnasium.make_vec, its wrappers and
    the SB3 shim rely on -- otherwise `object` (this image has no gymnasium)."""
    try:
        import gymnasium
    except ImportError:
        return object, object
    try:
        from gymnasium.experimental.vector import VectorEnv  # gymnasium ~= 0.29 (pyproject.toml:29)
    except ImportError:
        from gymnasium.vector import VectorEnv
    return gymnasium.Env, VectorEnv


_EnvBase, _VectorEnvBase = _gymnasium_bases()


class _Initializer:
    """Uniform states in `ends` (state_initializer.py:30-71, vectorised and seedable)."""

    def __init__(self, ends, seed):
        self._ends = ends
        self._generator = np.random.Generator(np.random.PCG64DXSM(seed))

    def initialize(self, num_envs):
        return self._generator.uniform(self._ends[0], self._ends[1], size=(num_envs, 2)).astype(np.float32)

    def propose(self, num_envs):
        """The rows initialize(num_envs) WOULD return, without consuming them: the device-resident step hands row r to the
        r-th environment that ends and initialize(k) then draws exactly the k rows that were used (same consumption as
        vector_environment.py:144).  The generator's state is saved and put back -- a deepcopy of the Generator cost 15-65 us
        per step, a quarter of a small environment's step -- and the rows are low + (high - low) * random(), which is what
        Generator.uniform computes (tests/test_harness_logic.py compares them bit for bit)."""
        bit_generator = self._generator.bit_generator
        state = bit_generator.state
        lo, hi = self._ends
        rows = (lo + (hi - lo) * self._generator.random((num_envs, 2))).astype(np.float32)
        bit_generator.state = state
        return rows


class _Ender:
    """TimeLimitEnder | DivergingEnder (episode_ender.py:580-656, :106-207, :369-452);
    max_steps None leaves only the diverging rule (the non-vector DiscreteSteps)."""

    def __init__(self, num_envs, max_steps, threshold, early_end_steps):
        self._num_envs = num_envs
        self._max_steps = max_steps
        self._threshold = threshold
        self._early_end_steps = early_end_steps
        self._steps = np.zeros(num_envs, dtype=np.int32)
        self._diverging_steps = np.zeros(num_envs, dtype=np.int32)
        self._last_diff = np.zeros(num_envs, dtype=np.float32)

    def step(self, states):
        self._steps += 1
        diff = abs(states[:, TARGET] - states[:, FOCUS])
        self._diverging_steps[diff > self._last_diff + self._threshold] += 1
        self._last_diff = diff

    def is_terminated(self):
        return np.full(self._num_envs, False)

    def is_truncated(self):
        truncated = self._diverging_steps >= self._early_end_steps
        if self._max_steps is not None:
            truncated = (self._steps >= self._max_steps) | truncated
        return truncated

    def reset(self, states, indices=None):
        if indices is None:
            indices = np.full(self._num_envs, True)
        self._steps[indices] = 0
        self._diverging_steps[indices] = 0
        self._last_diff[indices] = abs(states[:, TARGET] - states[:, FOCUS])

    def status(self, index):
        """episode_ender.py:646-656, :191-207, :439-452: what the visualiser prints."""
        diverging = self._diverging_steps[index]
        r_status = f"diverging {diverging} / {self._early_end_steps}" if diverging > 0 else ""
        if self._max_steps is None:
            return r_status
        l_status = f"step {self._steps[index]} / {self._max_steps}"
        return l_status + (", " if l_status and r_status else "") + r_status


def delta_bounds(lows, highs, max_change=None, include_original=False):
    """Observation bounds of DeltaObserver (state_observer.py:166-230): a change is bounded by
    high - low of what it is a change of, or by max_change where that is given (not NaN); with
    include_original the wrapped bounds come first.  float32, as gymnasium's Box."""
    lows = np.asarray(lows, dtype=np.float32)
    highs = np.asarray(highs, dtype=np.float32)
    diff = highs - lows
    if max_change is not None:
        max_change = np.asarray(max_change, dtype=np.float32)
        diff = np.where(np.isnan(max_change), diff, max_change).astype(np.float32)
    if include_original:
        return np.append(lows, -diff), np.append(highs, diff)
    return -diff, diff


def normaliser_from_bounds(low, high):
    """mid / scale of NormalizedObserver (state_observer.py:440-470): observations are mapped
    from [low, high] to [-1, 1] as clip((x - mid) / scale, -1, 1); everything float32."""
    spans = np.vstack([low, high]).astype(np.float32)
    return np.average(spans, axis=0), np.diff(spans / 2, axis=0).reshape(spans.shape[1])


def normaliser_constants(ends, max_move, min_focus, max_focus):
    """mid / scale of NormalizedObserver over DeltaObserver([IndexedElement, Focus], True,
    [max_move, nan]) -- custom_environments.py:196-218."""
    low, high = delta_bounds([ends[0], min_focus], [ends[1], max_focus], [max_move, np.nan], True)
    return normaliser_from_bounds(low, high)


class _Observer:
    """NormalizedObserver(DeltaObserver([IndexedElementObserver, FocusObserver], True,
    [max_move, nan])) -- state_observer.py:166-292, :386-517."""

    def __init__(self, num_envs, ends, max_move, focus_observer):
        self._focus = focus_observer
        self._mid, self._scale = normaliser_constants(
            ends, max_move, focus_observer.single_observation_space.low[0],
            focus_observer.single_observation_space.high[0])
        self._old = np.full((num_envs, 2), np.nan, dtype=np.float32)
        self._num_envs = num_envs
        self.single_observation_space = spaces.Box(-np.ones(4, dtype=np.float32), np.ones(4, dtype=np.float32),
                                                   dtype=np.float32)
        self.observation_space = spaces.batch_space(self.single_observation_space, num_envs)

    def _wrapped(self, states, indices):
        position = states[:, FOCUS].reshape((indices.sum(), 1))
        return np.hstack([position, self._focus.observe(states, indices)], dtype=np.float32)

    def _normalize(self, values):
        return np.clip((values - self._mid) / self._scale, -1, 1, dtype=np.float32)

    def observe(self, states, indices=None):
        if indices is None:
            indices = np.full(self._num_envs, True)
        wrapped = self._wrapped(states, indices)
        observations = np.hstack([wrapped, wrapped - self._old[indices]], dtype=np.float32)
        self._old[indices] = wrapped
        return self._normalize(observations)

    def reset(self, states, indices=None):
        if indices is None:
            indices = np.full(self._num_envs, True)
        wrapped = self._wrapped(states, indices)
        observations = np.hstack([wrapped, np.zeros(wrapped.shape, dtype=np.float32)], dtype=np.float32)
        self._old[indices] = wrapped
        return self._normalize(observations)


class _Rewarder:
    """DeltaRewarder + ObservationRewarder + OnTargetRewarder
    (episode_rewarder.py:86-155, :210-292)."""

    def __init__(self, scale, span, focus_value_o_index=1):
        self._scale = scale
        self._span = span
        self._o_index = focus_value_o_index
        self._old_states = None

    def reset(self, states, observations, indices=None):
        if self._old_states is not None and indices is not None:
            self._old_states[indices] = states[:, FOCUS]
        else:
            self._old_states = states[:, FOCUS]

    def reward(self, states, observations):
        moved = abs(states[:, FOCUS] - self._old_states) * -1.0 / self._scale
        self._old_states = states[:, FOCUS]
        on_target = (abs(states[:, TARGET] - states[:, FOCUS]) < self._span) * 1.0 + 0.0
        return (moved + observations[:, self._o_index]) + on_target


class _HostGlue:
    """The DiscreteSteps task with the reference's numpy glue on the host around the GPU
    render + focus (FocusObserver): everything VectorDiscreteSteps, DiscreteSteps and
    ContinuousJumps share.  Not an environment class by itself."""

    metadata = {"render_modes": ["rgb_array"], "render_fps": 4}

    def __init__(self, max_episode_steps=20, num_envs=1, render_mode=None, *, frame_height=300,
                 samples_per_pixel=100, seed=None, device=None, first_state_index=0, host_frames=False,
                 _diverging_only=False):
        """host_frames=True: the literal drop-in route (FastRenderer(host_frames=True): frames come back to the
        host every render and vision.focus_values uploads them again), identical results."""
        super().__init__()
        ends = (5.0, 10.0)
        target_radius = 0.25
        max_move = 5.0
        moves = max_move / 2.0 ** np.arange(6)

        assert render_mode is None or render_mode in self.metadata["render_modes"]
        self.render_mode = render_mode
        self.num_envs = num_envs

        self._renderer = render.FastRenderer(samples_per_pixel=samples_per_pixel, device=device,
                                             first_state_index=first_state_index, host_frames=host_frames)
        self._ender = _Ender(num_envs, None if _diverging_only else max_episode_steps, target_radius / 2, 3)
        self._initializer = _Initializer(ends, seed)
        self._focus_observer = state_observer.FocusObserver(num_envs, TARGET, FOCUS, ends, self._renderer,
                                                            frame_height)
        self._observer = _Observer(num_envs, ends, max_move, self._focus_observer)
        self._rewarder = _Rewarder(target_radius * 2, target_radius)
        # DiscreteMoveTransformer (state_transformer.py:222-266)
        self._action_set = np.concatenate([-moves, [0], moves[::-1]])
        self._limits = ends
        # custom_environments.py:229-238; observation element 1 is the focus value
        self._visualizer = episode_visualizer.HistoryVisualizer(
            num_envs, TARGET, FOCUS, 1, self._renderer, ends, ender=self._ender, target_radius=target_radius)

        self.single_action_space = spaces.Discrete(len(self._action_set))
        self.action_space = spaces.batch_space(self.single_action_space, num_envs)
        self.single_observation_space = self._observer.single_observation_space
        self.observation_space = self._observer.observation_space
        self._state = None

    # -- vector_environment.py:75-102 ------------------------------------------------------
    def reset(self, *, seed=None, options=None, state=None):
        """`state` (extension) pins the initial state instead of drawing it."""
        if seed is not None:
            self._initializer = _Initializer(self._limits, seed)
        self._state = (self._initializer.initialize(self.num_envs) if state is None
                       else np.array(state, dtype=np.float32).reshape(self.num_envs, 2))
        self._ender.reset(self._state)
        observations = self._observer.reset(self._state, None)
        self._rewarder.reset(self._state, observations)
        if self.render_mode == "rgb_array":
            self._visualizer.reset(self._state, observations)
        return observations, {}

    def _transform(self, states, actions):
        new_states = states.copy()
        new_states[:, FOCUS] += self._action_set[np.asarray(actions).flatten()]
        return np.clip(new_states, *self._limits)

    # -- vector_environment.py:104-164 -------------------------------------------------------
    def step(self, actions):
        assert self._state is not None
        self._state = self._transform(self._state, actions)
        self._ender.step(self._state)
        observations = self._observer.observe(self._state)
        rewards = self._rewarder.reward(self._state, observations)
        terminated = self._ender.is_terminated()
        truncated = self._ender.is_truncated()
        done = terminated | truncated
        if done.any():
            new_state = self._initializer.initialize(done.sum())
            self._state[done] = new_state
            self._ender.reset(new_state, done)
            new_observations = self._observer.reset(new_state, done)
            observations[done] = new_observations
            self._rewarder.reset(new_state, new_observations, done)
            if self.render_mode == "rgb_array":
                self._visualizer.reset(new_state, new_observations, done)
        if self.render_mode == "rgb_array":
            not_done = ~done
            self._visualizer.step(self._state[not_done], observations[not_done], not_done)
        return observations, rewards, terminated, truncated, {}

    def render(self):
        """vector_environment.py:166-176 -> HistoryVisualizer.visualize
        (episode_visualizer.py:188-201): the 600 px rendering of every environment (which
        advances / re-seeds the RNG states exactly as the reference's does) next to its
        performance plot."""
        if self.render_mode == "rgb_array":
            return self._visualizer.visualize()
        return None

    def render_frames(self):
        """Only the left halves of render(): uint8[num_envs, 600, 600, 3], no matplotlib."""
        return np.asarray(self._renderer.render(600))

    def close(self):
        self._renderer.close()


class VectorDiscreteSteps(_HostGlue, _VectorEnvBase):
    """The DiscreteSteps-v0 vector environment (custom_environments.py:114-241) with the
    reference's numpy glue on the host.  Same constructor arguments and defaults as the
    reference; frame_height / samples_per_pixel / seed / device / first_state_index are
    extensions (defaults = the reference's 300 px, 100 spp, device LOCAL_RANK).
    DeviceVectorDiscreteSteps is the same environment with the glue on the GPU (the default of
    registration.make_vec); results are identical bit for bit."""


class DiscreteSteps(_HostGlue, _EnvBase):
    """The single-environment DiscreteSteps (custom_environments.py:16-111 on
    environments/environment.py): one env, DivergingEnder only, unbatched returns."""

    def __init__(self, render_mode=None, **kwargs):
        super().__init__(num_envs=1, render_mode=render_mode, _diverging_only=True, **kwargs)
        self.action_space = self.single_action_space
        self.observation_space = self.single_observation_space

    def reset(self, *, seed=None, options=None, state=None):
        observations, info = super().reset(seed=seed, options=options, state=state)
        return observations[0], info

    def step(self, action):
        # environment.py: no auto-reset in the single-env shell
        self._state = self._transform(self._state, np.array([action]))
        self._ender.step(self._state)
        observations = self._observer.observe(self._state)
        if self.render_mode == "rgb_array":  # environment.py:119-120
            self._visualizer.step(self._state, observations)
        reward = self._rewarder.reward(self._state, observations)[0]
        return observations[0], reward, self._ender.is_terminated()[0], self._ender.is_truncated()[0], {}


class ContinuousJumps(_HostGlue, _EnvBase):
    """The single-environment ContinuousJumps (examples/custom_environments.py:244-339):
    one continuous action in [-1, 1] jumps the focus plane to the proportional position in
    [5, 10] unless the jump is shorter than target_radius / 2
    (ContinuousJumpTransformer, state_transformer.py:66-118); DivergingEnder only; reward =
    focus value + [stopped] * [on target] (ObservationRewarder + StoppedRewarder *
    OnTargetRewarder, episode_rewarder.py:210-292, :361-429)."""

    def __init__(self, render_mode=None, **kwargs):
        super().__init__(num_envs=1, render_mode=render_mode, _diverging_only=True, **kwargs)
        self._stop_threshold = abs(0.25 / 2.0)
        self.single_action_space = spaces.Box(-1, 1, dtype=np.float32)
        self.action_space = self.single_action_space
        self.observation_space = self.single_observation_space
        self._old_focus = None

    def reset(self, *, seed=None, options=None, state=None):
        observations, info = super().reset(seed=seed, options=options, state=state)
        self._old_focus = self._state[:, FOCUS]  # StoppedRewarder.reset keeps a view
        return observations[0], info

    def _transform(self, states, actions):
        new_states = states.copy()
        actions = (np.asarray(actions, dtype=np.float32).flatten() + 1) / 2.0
        moved_states = actions * (self._limits[1] - self._limits[0]) + self._limits[0]
        moved = abs(new_states[:, FOCUS] - moved_states) > self._stop_threshold
        new_states[moved, FOCUS] = moved_states[moved]
        return new_states

    def step(self, action):
        self._state = self._transform(self._state, np.array([action]))
        self._ender.step(self._state)
        observations = self._observer.observe(self._state)
        if self.render_mode == "rgb_array":
            self._visualizer.step(self._state, observations)
        stopped = (abs(self._state[:, FOCUS] - self._old_focus) < self._stop_threshold) * 1.0
        self._old_focus = self._state[:, FOCUS]
        on_target = (abs(self._state[:, TARGET] - self._state[:, FOCUS]) < 0.25) * 1.0 + 0.0
        reward = (observations[:, 1] + stopped * on_target)[0]
        return observations[0], reward, self._ender.is_terminated()[0], self._ender.is_truncated()[0], {}


class _DeviceShard:
    """One rf_ctx holding a contiguous range of device-resident DiscreteSteps environments
    (rf_env_*): the context, its RNG states at `first_state_index`, and the rf_env_config of the
    task (custom_environments.py:166-241).  DeviceVectorDiscreteSteps owns one,
    ShardedVectorDiscreteSteps one per device."""

    ENDS = (5.0, 10.0)
    TARGET_RADIUS = 0.25
    MAX_MOVE = 5.0
    EARLY_END_STEPS = 3  # DivergingEnder(..., early_end_steps=3), custom_environments.py:186-190

    def __init__(self, num_envs, max_episode_steps, frame_height, samples_per_pixel, device, first_state_index):
        import math

        from reinfocus_amd import _native, vision
        from reinfocus_amd.graphics import camera

        moves = self.MAX_MOVE / 2.0 ** np.arange(6)
        self.action_set = np.concatenate([-moves, [0], moves[::-1]])
        self.num_envs = num_envs
        self.frame_height = frame_height
        self.samples_per_pixel = samples_per_pixel
        self.first_state_index = int(first_state_index)
        self.max_episode_steps = max_episode_steps
        self.ctx = _native.Context(device)
        try:
            min_focus, max_focus = state_observer.cached_focus_extrema(self.ENDS, frame_height, samples_per_pixel,
                                                                       self.ctx.device)
            box = spaces.Box(min_focus, max_focus, dtype=np.float32)  # float32 rounding of the extrema
            mid, scale = normaliser_constants(self.ENDS, self.MAX_MOVE, box.low[0], box.high[0])
            cams = camera.FastCameras()
            cfg = _native.EnvConfig()
            cfg.n = num_envs
            cfg.n_actions = len(self.action_set)
            for i, a in enumerate(self.action_set):
                cfg.action_set[i] = float(a)
            cfg.limit_lo, cfg.limit_hi = self.ENDS
            cfg.max_steps = max_episode_steps if max_episode_steps else 0
            cfg.diverge_threshold = self.TARGET_RADIUS / 2
            cfg.early_end_steps = self.EARLY_END_STEPS
            for i in range(4):
                cfg.mid[i] = float(mid[i])
                cfg.scale[i] = float(scale[i])
            cfg.reward_scale = self.TARGET_RADIUS * 2
            cfg.on_target_span = self.TARGET_RADIUS
            cfg.half_width = cams._half_width
            cfg.half_height = cams._half_height
            cfg.tan_half_r = math.tan(math.radians(20 / 2))
            for i in range(3):
                cfg.look_from[i] = float(cams._look_from[i])
                cfg.cam_u[i] = float(cams._u[i])
                cfg.cam_v[i] = float(cams._v[i])
                cfg.cam_w[i] = float(cams._w[i])
            cfg.lens_radius = float(cams._half_aperture)
            cfg.frame_height = frame_height
            cfg.spp = samples_per_pixel
            cfg.gray_mode = vision.GRAY_MODE
            self.ctx.seed(num_envs * frame_height * frame_height, 0, self.first_state_index)
            self.ctx.env_configure(cfg)
        except Exception:
            self.ctx.close()
            raise

    # -- what HistoryVisualizer needs from a renderer / an ender (episode_visualizer.py:197, :268) --
    def render(self, frame_height):
        """FastRenderer.render on the renderer the reference's FocusObserver and visualiser share
        (render.py:165-188, :248-257): the scene set uploaded last, states re-created from seed 0
        when more are needed than exist."""
        needed = self.ctx.env_scene_len() * frame_height * frame_height
        if self.ctx.num_states() < needed:
            self.ctx.seed(needed, 0, self.first_state_index)
        return self.ctx.env_render(frame_height, self.samples_per_pixel)

    def status(self, index):
        """_Ender.status from the counters on the device."""
        steps, diverging = self.ctx.env_counters()
        r_status = f"diverging {diverging[index]} / {self.EARLY_END_STEPS}" if diverging[index] > 0 else ""
        if not self.max_episode_steps:
            return r_status
        l_status = f"step {steps[index]} / {self.max_episode_steps}"
        return l_status + (", " if l_status and r_status else "") + r_status


def _device_spaces(env, action_set, num_envs):
    env.single_action_space = spaces.Discrete(len(action_set))
    env.action_space = spaces.batch_space(env.single_action_space, num_envs)
    env.single_observation_space = spaces.Box(-np.ones(4, dtype=np.float32), np.ones(4, dtype=np.float32),
                                              dtype=np.float32)
    env.observation_space = spaces.batch_space(env.single_observation_space, num_envs)


class DeviceVectorDiscreteSteps(_VectorEnvBase):
    """VectorDiscreteSteps with the whole step resident on the GPU (rf_env_*, SURVEY.md
    section 8(f) item 1): same constructor, same reset/step results bit for bit, but a step
    only uploads the actions and the initializer's candidate states and downloads
    observations, rewards and flags.  This is what `DiscreteSteps-v0`'s vector entry point
    builds.  The initializer stays on the host (numpy PCG64DXSM): a copy of the generator
    proposes num_envs candidate states per step, the device hands row r to the r-th environment
    that ended, and the real generator then draws exactly the rows that were used -- the same
    consumption as VectorDiscreteSteps.  render_mode="rgb_array" works as in the reference
    (HistoryVisualizer on the environment's own renderer state: the 600 px render advances /
    re-seeds the RNG states the next step uses)."""

    metadata = {"render_modes": ["rgb_array"], "render_fps": 4}

    def __init__(self, max_episode_steps=20, num_envs=1, render_mode=None, *, frame_height=300,
                 samples_per_pixel=100, seed=None, device=None, first_state_index=0):
        super().__init__()
        assert render_mode is None or render_mode in self.metadata["render_modes"]
        self.render_mode = render_mode
        self.num_envs = num_envs
        self._shard = _DeviceShard(num_envs, max_episode_steps, frame_height, samples_per_pixel, device,
                                   first_state_index)
        self._ctx = self._shard.ctx
        self._limits = _DeviceShard.ENDS
        self._initializer = _Initializer(self._limits, seed)
        self._action_set = self._shard.action_set
        _device_spaces(self, self._action_set, num_envs)
        self._visualizer = None
        if render_mode == "rgb_array":  # custom_environments.py:229-238
            self._visualizer = episode_visualizer.HistoryVisualizer(
                num_envs, TARGET, FOCUS, 1, self._shard, self._limits, ender=self._shard,
                target_radius=_DeviceShard.TARGET_RADIUS)

    @property
    def _state(self):
        return self._ctx.env_states()

    def reset(self, *, seed=None, options=None, state=None):
        if seed is not None:
            self._initializer = _Initializer(self._limits, seed)
        initial = (self._initializer.initialize(self.num_envs) if state is None
                   else np.array(state, dtype=np.float32).reshape(self.num_envs, 2))
        observations = self._ctx.env_reset(initial)
        if self._visualizer is not None:
            self._visualizer.reset(initial, observations)
        return observations, {}

    def step(self, actions):
        pool = self._initializer.propose(self.num_envs)
        observations, rewards, truncated, used = self._ctx.env_step(actions, pool)
        if used:
            self._initializer.initialize(used)  # consume exactly the rows that were used
        if self._visualizer is not None:  # vector_environment.py:149-156
            state = self._state
            if used:
                self._visualizer.reset(state[truncated], observations[truncated], truncated)
            self._visualizer.step(state[~truncated], observations[~truncated], ~truncated)
        return observations, rewards, np.full(self.num_envs, False), truncated, {}

    def render(self):
        """vector_environment.py:166-176."""
        if self._visualizer is not None:
            return self._visualizer.visualize()
        return None

    def render_frames(self):
        """Only the left halves of render(): the 600 px frames of the scene set uploaded last."""
        return self._shard.render(episode_visualizer.HistoryVisualizer.FRAME)

    def close(self):
        self._ctx.close()


def split_environments(num_envs, shards):
    """Contiguous env ranges [first, first + count) per shard, as even as possible."""
    counts = [num_envs // shards + (1 if g < num_envs % shards else 0) for g in range(shards)]
    firsts = np.concatenate([[0], np.cumsum(counts)[:-1]]).astype(int)
    return [(int(f), int(c)) for f, c in zip(firsts, counts)]


class _ShardSet:
    """What HistoryVisualizer needs from "the renderer" and "the ender" of a sharded environment
    (episode_visualizer.py:197, :268): every shard draws the scene set it uploaded last -- each
    behaves like the reference's one shared renderer for its own environment range -- and the rows
    are stacked in shard order."""

    def __init__(self, env):
        self._env = env

    def render(self, frame_height):
        # After a step in which k > 0 environments ended, one device holds the k compacted rows of the auto-reset
        # (rf_env_scene_len): here those are the shards that had resets, in shard order -- global index order, the
        # compacted order.  A shard without resets that step still holds its full set and contributes nothing then.
        ended = self._env._last_ended
        partial = ended is not None and sum(ended) > 0
        chosen = [g for g in range(len(self._env._shards)) if not partial or ended[g] > 0]
        futures = [self._env._threads[g].submit(lambda shard=self._env._shards[g]: np.asarray(shard.render(frame_height)))
                   for g in chosen]
        return np.concatenate([f.result() for f in futures])

    def status(self, index):
        g, local = self._env._locate(index)
        return self._env._threads[g].submit(self._env._shards[g].status, local).result()


class ShardedVectorDiscreteSteps(_VectorEnvBase):
    """DeviceVectorDiscreteSteps over several GPUs of one node (the
    reference has no counterpart: vector_environment.py:104-164 steps all environments on one
    device).  Environments are independent, so device g owns the contiguous range
    [first_g, first_g + n_g) -- one rf_ctx and one host thread per device (a single-worker executor
    each; ctypes releases the GIL), no device-to-device traffic, the host concatenates observations /
    rewards / flags (8 + 16 + 1 bytes per environment).

    RNG states: shard g is seeded at global state index first_state_index + first_g * h * h
    (pixel index = e * h * w + y * w + x, render.py:217), so full renders draw exactly what one
    device holding all environments would draw.  Initializer: one generator for the whole
    environment; the r-th environment that ended, in global index order, takes the r-th drawn
    state, as on one device -- which is why a step has two halves: the rows a shard takes depend on
    how many environments ended before it.  The cut is before the render (rf_env_step_plan /
    rf_env_step_run: which environments end depends on their counters alone), so the first half is
    cheap and the second is the fused step -- one render launch per shard.

    Auto-reset renders.  On one device the partial render indexes RNG states from 0 over the
    compacted rows of all environments that ended (vector_environment.py:144 -> render.py:217):
    compacted row r draws from the states of environment slot r.
     * default: every shard renders its own ended environments from its own state base -- balanced,
       no traffic, but after the first auto-reset a sharded run and a one-device run are different
       (equally valid) sample paths;
     * exact=True: row r is rendered by the shard that owns slot r (rf_env_render_states) and the
       focus value returns through the host to the shard the environment lives on
       (rf_env_step_end_given): bit-identical to one device through any number of auto-resets, at the
       price of the first shards rendering everybody's resets.  Meant for tests and reproducibility
       studies, not for throughput.
    render_mode="rgb_array": HistoryVisualizer over the shards (each shard's 600 px render advances /
    re-seeds that shard's RNG states as the reference's single renderer would for its range; the
    exact mode does not extend to visualised runs).  `devices` may name a device more than once
    (several contexts on one GPU: tests, rehearsals).  numa_pin: every shard thread restricts itself
    to the CPUs of its GPU's NUMA node (`placements` says where each shard ended up)."""

    metadata = {"render_modes": ["rgb_array"], "render_fps": 4}

    def __init__(self, max_episode_steps=20, num_envs=1, render_mode=None, *, devices=None, frame_height=300,
                 samples_per_pixel=100, seed=None, first_state_index=0, exact=False, numa_pin=True):
        import concurrent.futures
        from reinfocus_amd import _native

        super().__init__()
        assert render_mode is None or render_mode in self.metadata["render_modes"]
        # (one array of RNG states aliased by 300 px and 600 px renders cannot be reproduced across devices, and the
        # exact mode's row renders replace the scene sets the visualiser would draw)
        assert not (exact and render_mode), "exact=True does not extend to visualised runs (render_mode)"
        self.render_mode = render_mode
        self.exact = bool(exact)
        self._last_ended = None  # environments that ended in the last step, per shard (None: after a reset)
        if devices is None:
            devices = list(range(_native.device_count()))
        devices = [int(d) for d in devices]
        assert 1 <= len(devices) <= num_envs, "need between 1 and num_envs devices"
        self.num_envs = num_envs
        self.devices = devices
        self._ranges = split_environments(num_envs, len(devices))
        self._limits = _DeviceShard.ENDS
        self._initializer = _Initializer(self._limits, seed)
        # one thread per shard for the life of the environment: a shard's context is only ever
        # touched from its own thread
        self._threads = [concurrent.futures.ThreadPoolExecutor(max_workers=1, thread_name_prefix=f"reinfocus-shard{g}")
                         for g in range(len(devices))]
        pixels = frame_height * frame_height
        self._shards = []
        try:
            def make_shard(count, device, first):
                # (in the shard's own thread, before its context exists: the thread that will drive the GPU -- and
                # first-touch its pinned staging buffers -- runs on the CPUs of that GPU's NUMA node)
                placement = _native.device_info(device)
                placement["cpus"] = _native.pin_to_numa_node(placement["numa_node"]) if numa_pin else None
                shard = _DeviceShard(count, max_episode_steps, frame_height, samples_per_pixel, device,
                                     first_state_index + first * pixels)
                shard.placement = placement
                return shard

            futures = [thread.submit(make_shard, count, device, first)
                       for thread, device, (first, count) in zip(self._threads, devices, self._ranges)]
            for future in futures:
                try:
                    self._shards.append(future.result())
                except Exception:
                    for thread, other in zip(self._threads, futures):
                        try:  # (a context is only ever touched from its own thread)
                            thread.submit(other.result().ctx.close).result()
                        except Exception:
                            pass
                    raise
        except Exception:
            for thread in self._threads:
                thread.shutdown(wait=True)
            raise
        self._action_set = self._shards[0].action_set
        _device_spaces(self, self._action_set, num_envs)
        self._visualizer = None
        if render_mode == "rgb_array":  # custom_environments.py:229-238
            both = _ShardSet(self)
            self._visualizer = episode_visualizer.HistoryVisualizer(
                num_envs, TARGET, FOCUS, 1, both, self._limits, ender=both, target_radius=_DeviceShard.TARGET_RADIUS)

    @property
    def placements(self):
        """Per shard: {"device", "pci_bus_id", "numa_node", "cpus"} -- which physical GPU the shard's context is on
        and the CPUs its host thread was restricted to (None: not pinned)."""
        return [dict(shard.placement) for shard in self._shards]

    def _submit(self, function, *per_shard):
        return [thread.submit(function, shard, *(a[g] for a in per_shard))
                for g, (thread, shard) in enumerate(zip(self._threads, self._shards))]

    def _each(self, function, *per_shard):
        """function(shard, *args_g) on every shard's own thread; results in shard order."""
        return [f.result() for f in self._submit(function, *per_shard)]

    @staticmethod
    def _results(futures):
        """Every future's result; all of them are waited for before the first error is raised (no shard is left
        running behind an exception)."""
        results, errors = [], []
        for future in futures:
            try:
                results.append(future.result())
            except Exception as error:  # noqa: BLE001 -- re-raised below
                errors.append(error)
        if errors:
            raise errors[0]
        return results

    def _slices(self, array):
        return [array[first:first + count] for first, count in self._ranges]

    def _locate(self, index):
        for g, (first, count) in enumerate(self._ranges):
            if first <= index < first + count:
                return g, index - first
        raise IndexError(index)

    @property
    def _state(self):
        return np.concatenate(self._each(lambda shard: shard.ctx.env_states()))

    def reset(self, *, seed=None, options=None, state=None):
        if seed is not None:
            self._initializer = _Initializer(self._limits, seed)
        initial = (self._initializer.initialize(self.num_envs) if state is None
                   else np.array(state, dtype=np.float32).reshape(self.num_envs, 2))
        observations = np.concatenate(self._each(lambda shard, rows: shard.ctx.env_reset(rows), self._slices(initial)))
        self._last_ended = None
        if self._visualizer is not None:
            self._visualizer.reset(initial, observations)
        return observations, {}

    def _begin(self, actions):
        """First half of the step on every shard -- rf_env_step_plan (transform, enders, ranking: the cut before the
        render that lets the second half run as ONE render launch), in the exact mode rf_env_step_begin (the cut after
        the full render: its row renders happen on other shards).  If any shard fails, the shards whose half did run
        drop it (rf_env_step_abort: they then insist on a reset) and the first error is raised."""
        if self.exact:
            futures = self._submit(lambda shard, a: shard.ctx.env_step_begin(a), self._slices(actions))
        else:
            futures = self._submit(lambda shard, a: (None, None, shard.ctx.env_step_plan(a)), self._slices(actions))
        results, errors = [], []
        for future in futures:
            try:
                results.append(future.result())
            except Exception as error:  # noqa: BLE001 -- re-raised below
                results.append(None)
                errors.append(error)
        if errors:
            aborts = [thread.submit(shard.ctx.env_step_abort)
                      for thread, shard, result in zip(self._threads, self._shards, results) if result is not None]
            for abort in aborts:
                abort.result()
            raise errors[0]
        return results

    def step(self, actions):
        actions = np.asarray(actions).reshape(self.num_envs)
        # every shard validates its slice again, but a bad action must not leave some shards half way
        # through a step: check all of them before any shard begins
        if actions.size and (actions.min() < 0 or actions.max() >= len(self._action_set)):
            raise AssertionError(f"action outside [0, {len(self._action_set)})")
        pool = self._initializer.propose(self.num_envs)
        firsts = self._begin(actions)
        ended = [k for _, _, k in firsts]
        starts = np.concatenate([[0], np.cumsum(ended)]).astype(int)
        total = int(starts[-1])
        rows = [pool[starts[g]:starts[g] + ended[g]] for g in range(len(self._shards))]
        try:
            if self.exact and total:
                # compacted row r is rendered where environment slot r's RNG states live
                spans = [(first, min(first + count, total)) for first, count in self._ranges]
                renders = [thread.submit(shard.ctx.env_render_states, pool[lo:hi])
                           for thread, shard, (lo, hi) in zip(self._threads, self._shards, spans) if lo < hi]
                focus = np.concatenate(self._results(renders))
                assert len(focus) == total
                values = [focus[starts[g]:starts[g] + ended[g]] for g in range(len(self._shards))]
                observations = self._results(self._submit(lambda shard, r, v: shard.ctx.env_step_end_given(r, v),
                                                          rows, values))
            elif self.exact:
                observations = self._results(self._submit(lambda shard, r: shard.ctx.env_step_end_given(r, np.zeros(0)),
                                                          rows))
            else:
                finished = self._results(self._submit(lambda shard, r: shard.ctx.env_step_run(r), rows))
                observations = [o for o, _, _ in finished]
                firsts = [(r, t, k) for (_, r, t), (_, _, k) in zip(finished, firsts)]
        except Exception:
            # some shard failed in the second half: the others must not keep a half-finished step (those that had
            # finished theirs refuse the abort, which is fine) -- every shard then insists on a reset or was done
            for abort in [thread.submit(shard.ctx.env_step_abort) for thread, shard in zip(self._threads, self._shards)]:
                try:
                    abort.result()
                except Exception:  # noqa: BLE001 -- no open step on that shard
                    pass
            raise
        self._last_ended = ended
        if total:
            self._initializer.initialize(total)  # consume exactly the rows that were used
        observations = np.concatenate(observations)
        rewards = np.concatenate([r for r, _, _ in firsts])
        truncated = np.concatenate([t for _, t, _ in firsts])
        if self._visualizer is not None:  # vector_environment.py:149-156
            state = self._state
            if total:
                self._visualizer.reset(state[truncated], observations[truncated], truncated)
            self._visualizer.step(state[~truncated], observations[~truncated], ~truncated)
        return observations, rewards, np.full(self.num_envs, False), truncated, {}

    def render(self):
        """vector_environment.py:166-176."""
        if self._visualizer is not None:
            return self._visualizer.visualize()
        return None

    def render_frames(self):
        """Only the left halves of render(): the 600 px frames every shard's renderer holds."""
        return _ShardSet(self).render(episode_visualizer.HistoryVisualizer.FRAME)

    def close(self):
        for thread, shard in zip(self._threads, self._shards):
            thread.submit(shard.ctx.close).result()
        self._shards = []
        for thread in self._threads:
            thread.shutdown(wait=True)
        self._threads = []
