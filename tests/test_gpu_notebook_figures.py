"""The figures the reference's notebooks hold as outputs, redrawn from the HIP path's frames and
compared pixel by pixel with the reference's (tests/notebook_figures.py; the CPU twin with the
oracle's frames is tests/test_reference_notebook_figures.py)."""

import numpy as np
import pytest

from tests import notebook_figures as nf

pytestmark = pytest.mark.gpu


def test_general_renderer_reproduces_render_notebook_figures():
    """examples/render.ipynb: five scenes (rectangles, spheres, both; cameras looking from the
    side) through render.render -> rf_render_general."""
    from reinfocus_amd.graphics import render

    worlds, cameras, frame_shape = nf.render_notebook_scenes()
    frames = render.render(worlds, cameras, frame_shape, device=0)  # 100 spp, the reference's default
    assert frames.shape == (5, 300, 600, 3) and frames.dtype == np.uint8
    for i, frame in enumerate(frames):
        differing, largest, total = nf.compare(nf.imshow_png(frame), f"render_cell3_{i}.png")
        assert total == 296 * 552
        assert differing <= 120 and largest <= 3, (i, differing, largest)


def test_fast_path_reproduces_environment_notebook_figures():
    """examples/environment.ipynb: one DiscreteSteps episode of 7 steps and 8 e.render() figures
    (600 px FastRenderer frame | plot of every focus position and focus value so far)."""
    from reinfocus_amd.environments import harness

    env = harness.DiscreteSteps(render_mode="rgb_array", device=0)
    names = []
    observations = []
    for name, image, observation in nf.episode(env):
        names.append(name)
        observations.append(observation)
        found = nf.compare_episode_figure(nf.show_png(image), name)
        assert found["frame"][0] <= 6 and found["frame"][1] <= 1, (name, found)
        assert found["plot"] <= 250 and found["digit_box"][0] <= 20 and found["digit_box"][1] <= 16, (name, found)
    assert names == nf.EPISODE_FIGURES
    assert repr(observations[1]) == "array([ 0.59203607, -0.873161  ,  0.0625    , -0.01416067], dtype=float32)"
    env.close()


def test_device_resident_environment_follows_the_same_episode():
    """The env id's device-resident vector environment (one member) through the same episode:
    the same observations and the same 600 px frames as the single-environment DiscreteSteps
    whose figures the test above compares with the reference's."""
    from reinfocus_amd import registration
    from reinfocus_amd.environments import harness

    single = harness.DiscreteSteps(render_mode="rgb_array", device=0)
    vector = registration.make_vec("DiscreteSteps-v0", num_envs=1, vector_kwargs={"render_mode": "rgb_array", "device": 0})
    assert type(vector) is harness.DeviceVectorDiscreteSteps

    class AsSingle:  # the notebook's calls on a one-member vector environment
        _state = property(lambda self: vector._state)

        def reset(self, state):
            observations, info = vector.reset(state=state)
            return observations[0], info

        def step(self, action):
            observations, rewards, terminated, truncated, info = vector.step(np.array([action]))
            return observations[0], rewards[0], terminated[0], truncated[0], info

        def render(self):
            return vector.render()

    for (name, image, observation), (_, image_v, observation_v) in zip(nf.episode(single), nf.episode(AsSingle())):
        assert np.array_equal(observation, observation_v), name
        assert np.array_equal(image[:, :600], image_v[:, :600]), name
    single.close()
    vector.close()
