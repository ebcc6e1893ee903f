"""Host-side visualiser pieces: the known answers of the
reference's tests/histories_test.py and tests/environments/episode_visualizer_test.py, the
bilinear resize that stands in for cv2.resize, and the compositing with a stub renderer."""

import numpy as np
import pytest
from numpy import testing

from reinfocus_amd import histories
from reinfocus_amd.environments import episode_visualizer


# ---- histories_test.py:16-92 ------------------------------------------------------------------
def test_new_histories_are_empty():
    testee = histories.Histories(3, 5)
    for i in range(3):
        testing.assert_allclose(testee.get_history(i), [])


def test_append_events():
    testee = histories.Histories(2, 3)
    for events in ([1, 4], [2, 3], [3, 2], [4, 1]):
        testee.append_events(events)
    testing.assert_allclose(testee.get_history(0), [2, 3, 4])
    testing.assert_allclose(testee.get_history(1), [3, 2, 1])


def test_partial_append_events():
    testee = histories.Histories(2, 3)
    testee.append_events([1, 4])
    testee.append_events([2, 3])
    testee.append_events([3], np.array([True, False]))
    testee.append_events([2], np.array([False, True]))
    testee.append_events([4, 1])
    testing.assert_allclose(testee.get_history(0), [2, 3, 4])
    testing.assert_allclose(testee.get_history(1), [3, 2, 1])


def test_most_recent_events():
    testee = histories.Histories(4, 2)
    testee.append_events([1, 2, 3, 4])
    testee.reset([False, False, True, False])
    testing.assert_allclose(testee.most_recent_events(), [1, 2, np.nan, 4])


def test_reset():
    testee = histories.Histories(3, 2)
    testee.append_events([1, 3, 5])
    testee.reset([True, False, False])
    testee.append_events([2, 4, 6])
    testee.reset([False, False, True])
    testing.assert_allclose(testee.get_history(0), [2])
    testing.assert_allclose(testee.get_history(1), [3, 4])
    testing.assert_allclose(testee.get_history(2), [])
    testee.append_events([1, 5, 1])
    testee.reset([True, False, True])
    testing.assert_allclose(testee.get_history(1), [4, 5])
    assert testee.data.dtype == np.float32


# ---- episode_visualizer_test.py:20-64 ---------------------------------------------------------
def _black_to_white():
    from matplotlib import colors

    return colors.LinearSegmentedColormap.from_list("", ["black", "white"])


def test_fade():
    testing.assert_allclose(episode_visualizer.fading_colours(_black_to_white(), 5, 3, p=1),
                            [(0.6,) * 4, (0.8,) * 4, (1.0,) * 4])


def test_high_power_fades_fast():
    lower = episode_visualizer.fading_colours(_black_to_white(), 5, 5, p=2)
    higher = episode_visualizer.fading_colours(_black_to_white(), 5, 5, p=1)
    testing.assert_allclose(lower[-1], higher[-1])
    testing.assert_array_less(lower[:-1], higher[:-1])


def test_high_power_increasingly_fades():
    differences = np.diff(episode_visualizer.fading_colours(_black_to_white(), 5, 5, p=3), axis=0)
    assert np.all(differences[1:] > differences[:-1])


# ---- resize ------------------------------------------------------------------------------------
def _resize_float(image, width, height):
    """Plain float bilinear with pixel-centre alignment and clamped taps."""
    src_h, src_w = image.shape[:2]
    fx = np.clip((np.arange(width) + 0.5) * src_w / width - 0.5, 0, src_w - 1)
    fy = np.clip((np.arange(height) + 0.5) * src_h / height - 0.5, 0, src_h - 1)
    x0, y0 = np.floor(fx).astype(int), np.floor(fy).astype(int)
    x1, y1 = np.minimum(x0 + 1, src_w - 1), np.minimum(y0 + 1, src_h - 1)
    wx, wy = (fx - x0)[None, :, None], (fy - y0)[:, None, None]
    image = image.astype(np.float64)
    top = image[y0][:, x0] * (1 - wx) + image[y0][:, x1] * wx
    bottom = image[y1][:, x0] * (1 - wx) + image[y1][:, x1] * wx
    return top * (1 - wy) + bottom * wy


def test_resize_linear_u8():
    rng = np.random.default_rng(3)
    image = rng.integers(0, 256, size=(48, 64, 3), dtype=np.uint8)
    same = episode_visualizer.resize_linear_u8(image, 64, 48)
    assert np.array_equal(same, image)
    up = episode_visualizer.resize_linear_u8(image, 80, 60)  # the 480x640 -> 600x800 ratio
    assert up.shape == (60, 80, 3) and up.dtype == np.uint8
    assert np.abs(up.astype(np.float64) - _resize_float(image, 80, 60)).max() <= 1.0
    down = episode_visualizer.resize_linear_u8(image, 32, 24)
    assert np.abs(down.astype(np.float64) - _resize_float(image, 32, 24)).max() <= 1.0
    flat = np.full((10, 12, 3), 201, dtype=np.uint8)
    assert np.all(episode_visualizer.resize_linear_u8(flat, 31, 17) == 201)


# ---- compositing with a stub renderer ----------------------------------------------------------
class _StubRenderer:
    def __init__(self, num_envs):
        self.calls = []
        self._num_envs = num_envs

    def render(self, frame_height):
        self.calls.append(frame_height)
        frames = np.zeros((self._num_envs, frame_height, frame_height, 3), dtype=np.uint8)
        frames[:, :, :, 1] = 77
        return frames


class _StubEnder:
    def status(self, index):
        return f"step {index} / 20"


def test_history_visualizer_composes_rows():
    pytest.importorskip("matplotlib")
    num_envs = 2
    renderer = _StubRenderer(num_envs)
    testee = episode_visualizer.HistoryVisualizer(num_envs, 0, 1, 1, renderer, (5.0, 10.0), ender=_StubEnder(),
                                                  history_length=4, target_radius=0.25)
    states = np.array([[6.0, 9.0], [7.0, 5.5]], dtype=np.float32)
    observations = np.array([[0.1, -0.8, 0, 0], [0.2, -0.5, 0, 0]], dtype=np.float32)
    testee.reset(states, observations)
    for move in range(3):
        states = states.copy()
        states[:, 1] -= 0.5
        observations = observations.copy()
        observations[:, 1] += 0.1
        testee.step(states, observations)
    # a partial reset touches only the selected environment
    testee.reset(np.array([[8.0, 6.0]], dtype=np.float32), np.array([[0.0, -0.9, 0, 0]], dtype=np.float32),
                 np.array([False, True]))
    testing.assert_allclose(testee._move_histories.get_history(0), [9.0, 8.5, 8.0, 7.5])
    testing.assert_allclose(testee._move_histories.get_history(1), [6.0])
    testing.assert_allclose(testee._focus_histories.get_history(1), [-0.9])
    assert list(testee._current_moves) == [3, 0]
    testing.assert_allclose(testee._targets, [6.0, 8.0])

    image = testee.visualize()
    assert renderer.calls == [600]
    assert image.dtype == np.uint8 and image.shape == (num_envs * 600, 600 + 800, 3)
    # left halves are the rendered frames, untouched; right halves are mostly white canvas
    assert np.all(image[:, :600, 1] == 77) and np.all(image[:, :600, 0] == 0)
    assert (image[:, 600:] == 255).mean() > 0.8
    assert not np.array_equal(image[:600, 600:], image[600:, 600:])


def test_ender_status_strings():
    from reinfocus_amd.environments import harness

    ender = harness._Ender(2, 20, 0.125, 3)
    states = np.array([[6.0, 9.0], [7.0, 5.5]], dtype=np.float32)
    ender.reset(states)
    assert ender.status(0) == "step 0 / 20"
    states[0, 1] = 10.0  # env 0 diverges by more than the threshold
    ender.step(states)
    assert ender.status(0) == "step 1 / 20, diverging 1 / 3"
    assert ender.status(1) == "step 1 / 20"
    assert harness._Ender(1, None, 0.125, 3).status(0) == ""
