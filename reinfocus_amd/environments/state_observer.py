"""FocusObserver: the caller of the render-and-measure hot path.

Mirrors reinfocus/environments/state_observer.py:62-97 (BaseObserver), :295-320
(cached_focus_extrema) and :323-383 (FocusObserver): same constructor arguments,
same observe/reset semantics and return shapes, so the reference's wrapper observers
(DeltaObserver, NormalizedObserver, ...) can hold it unchanged.  The render and the
focus measure run on the GPU; only 8 bytes per environment come back to the host.
"""

import functools

import numpy as np

from reinfocus_amd import vision
from reinfocus_amd.environments import spaces
from reinfocus_amd.graphics import render


class BaseObserver:
    """A state observer that produces observations within some range
    (state_observer.py:57-97)."""

    def __init__(self, num_envs, min_obs, max_obs):
        self.single_observation_space = spaces.Box(min_obs, max_obs, dtype=np.float32)
        self.observation_space = spaces.batch_space(self.single_observation_space, num_envs)

    def observe(self, states, indices=None):
        raise NotImplementedError

    def reset(self, states, indices=None):
        if indices is None:
            indices = np.full(self.observation_space.shape[0], True)
        return self.observe(states, indices)


@functools.lru_cache(maxsize=None)
def _focus_extrema(ends, frame_height, samples_per_pixel, device):
    max_targets = np.linspace(*ends, 11)
    renderer = render.FastRenderer(samples_per_pixel=samples_per_pixel, device=device)
    try:
        renderer.update_targets(np.append(ends, max_targets))
        renderer.update_focus_planes(np.append(ends[::-1], max_targets))
        focus_values = vision.focus_values(renderer.render(frame_height))
    finally:
        renderer.close()  # the fresh renderer of state_observer.py:314 is garbage afterwards
    return min(focus_values[0:2]), max(focus_values[2:13])


def cached_focus_extrema(ends, frame_height, samples_per_pixel=100, device=None):
    """state_observer.py:295-320: the least focus (target and focus plane at opposite
    ends) and the greatest (both at the same place, 11 places), from one render of 13
    environments by a FRESH FastRenderer (seed-0 states), cached per argument set.
    samples_per_pixel is an extension (the reference always uses FastRenderer()'s 100)."""
    return _focus_extrema((float(ends[0]), float(ends[1])), int(frame_height), int(samples_per_pixel), device)


class FocusObserver(BaseObserver):
    """Observes the focus value of each environment's rendered scene
    (state_observer.py:323-383)."""

    def __init__(self, num_envs, target_index, focus_plane_index, ends, renderer, frame_height=300):
        min_focus, max_focus = cached_focus_extrema(
            ends, frame_height, renderer._samples_per_pixel, renderer._ctx.device
        )
        super().__init__(num_envs, min_focus, max_focus)
        self._target_index = target_index
        self._focus_plane_index = focus_plane_index
        self._renderer = renderer
        self._frame_height = frame_height

    def observe(self, states, indices=None):
        """state_observer.py:359-383: every row of `states` is rendered and scored;
        `indices` only sizes the result (k = indices.sum() rows on a partial reset)."""
        if indices is None:
            indices = np.full(self.observation_space.shape[0], True)
        self._renderer.update_targets(states[:, self._target_index])
        self._renderer.update_focus_planes(states[:, self._focus_plane_index])
        return np.reshape(
            vision.focus_values(self._renderer.render(self._frame_height)),
            (indices.sum(), self.observation_space.shape[1]),
        )
