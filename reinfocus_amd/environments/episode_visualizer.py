"""Human-targeted visualisation of running episodes.

Mirror of reinfocus/environments/episode_visualizer.py: `fading_colours` (:19-39) and
`HistoryVisualizer` (:84-301).  The only GPU work is `renderer.render(600)`
(episode_visualizer.py:197): it runs on the MI355X path and -- as in the reference -- advances
the renderer's RNG states (and re-creates them the first time 600 px is larger than anything
rendered before, render.py:256-257), so an environment that is visualised keeps producing
the observations the reference would produce.  The frames come back as a lazy DeviceFrames and
are copied to the host once, for the compositing.

The graphs are drawn with matplotlib exactly as the reference draws them; cv2.resize /
hconcat / vconcat are replaced by numpy (`resize_linear_u8` restates OpenCV's 8-bit
fixed-point INTER_LINEAR; there is no cv2 in this image to compare it with, and the graph
pixels depend on the matplotlib version anyway, so only the left halves -- the rendered
frames -- are claimed to be bit-identical to the reference).
"""

import numpy as np

from reinfocus_amd import histories


def fading_colours(cmap, max_n, n, p=2):
    """n RGBA colours fading from cmap(1) towards cmap((1 / max_n) ** p); the alpha channel
    fades with them (episode_visualizer.py:19-39)."""
    samples = np.linspace(1 - (n - 1) / max_n, 1, n) ** p
    colours = cmap(samples)
    colours[:, -1] = samples
    return colours


def resize_linear_u8(image, width, height):
    """cv2.resize(image, (width, height)) for uint8 HxWxC images, default INTER_LINEAR:
    pixel centres aligned (src = (dst + 0.5) * scale - 0.5), taps clamped at the borders,
    11-bit fixed-point weights, horizontal pass first, rounding `(... + 2) >> 2`."""
    image = np.asarray(image, dtype=np.uint8)
    src_h, src_w = image.shape[:2]

    def taps(dst_n, src_n):
        f = (np.arange(dst_n) + 0.5) * (src_n / dst_n) - 0.5
        i0 = np.floor(f).astype(np.int64)
        w1 = f - i0
        w1[i0 < 0] = 0.0
        i0[i0 < 0] = 0
        over = i0 >= src_n - 1
        w1[over] = 0.0
        i0[over] = src_n - 1
        i1 = np.minimum(i0 + 1, src_n - 1)
        b1 = np.rint(w1 * 2048).astype(np.int64)
        return i0, i1, 2048 - b1, b1

    x0, x1, ax0, ax1 = taps(width, src_w)
    y0, y1, by0, by1 = taps(height, src_h)
    rows = image.astype(np.int64)
    horizontal = rows[:, x0] * ax0[None, :, None] + rows[:, x1] * ax1[None, :, None]
    top, bottom = horizontal[y0] >> 4, horizontal[y1] >> 4
    out = (((top * by0[:, None, None]) >> 16) + ((bottom * by1[:, None, None]) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


class HistoryVisualizer:
    """Stacks one row per environment: the 600 px rendering on the left, a plot of the last
    `history_length` focus positions / focus values on the right
    (episode_visualizer.py:84-301; same constructor arguments)."""

    def __init__(self, num_envs, target_index, focus_plane_index, focus_value_index, renderer, limits,
                 ender=None, history_length=10, target_radius=None):
        self._num_envs = num_envs
        self._target_index = target_index
        self._focus_plane_index = focus_plane_index
        self._focus_value_index = focus_value_index
        self._limits = limits
        self._history_length = history_length
        self._target_radius = target_radius
        self._ender = ender
        self._renderer = renderer

        self._current_moves = np.zeros(num_envs, dtype=np.int32)
        self._targets = np.zeros(num_envs, dtype=np.float32)
        self._move_histories = histories.Histories(num_envs, history_length)
        self._focus_histories = histories.Histories(num_envs, history_length)

    def step(self, states, observations, indices=None):
        """One timestep of the selected environments (episode_visualizer.py:133-155)."""
        if indices is None:
            indices = np.full(self._num_envs, True)
        self._current_moves[indices] += 1
        self._move_histories.append_events(states[:, self._focus_plane_index], indices)
        self._focus_histories.append_events(observations[:, self._focus_value_index], indices)

    def reset(self, states, observations, indices=None):
        """The selected environments started new episodes (episode_visualizer.py:157-185)."""
        if indices is None:
            indices = np.full(self._num_envs, True)
        self._current_moves[indices] = 0
        self._targets[indices] = states[:, self._target_index]
        self._move_histories.reset(indices)
        self._move_histories.append_events(states[:, self._focus_plane_index], indices)
        self._focus_histories.reset(indices)
        self._focus_histories.append_events(observations[:, self._focus_value_index], indices)

    def visualize(self):
        """uint8[num_envs * 600, 600 + graph width, 3] (episode_visualizer.py:188-201)."""
        renderings = np.asarray(self._renderer.render(600))
        graphs = [self._visualize_single_history(i) for i in range(self._num_envs)]
        return np.concatenate([np.concatenate([r, g], axis=1) for r, g in zip(renderings, graphs)], axis=0)

    def _visualize_single_history(self, env_index, frame_height=600):
        """The performance plot of one environment (episode_visualizer.py:203-301)."""
        import matplotlib

        matplotlib.use("Agg", force=False)
        from matplotlib import pyplot

        focus_history = self._focus_histories.get_history(env_index)
        move_history = self._move_histories.get_history(env_index)
        target = self._targets[env_index]
        n_focus_history = len(focus_history)

        figure, axes = pyplot.subplots()
        axes.set_xlim(*self._limits)
        axes.set_ylim(-1.0, 1.0)
        x_label = f"focus position {self._current_moves[env_index]}\n"
        if self._ender is not None:
            x_label += self._ender.status(env_index)
        axes.set_xlabel(x_label)
        axes.set_ylabel("focus value")
        axes.axvline(x=target, linestyle=":", color="darkorange", label="target")
        if self._target_radius is not None and self._target_radius > 0.0:
            axes.axvspan(target - self._target_radius, target + self._target_radius, edgecolor="darkorange",
                         facecolor=("darkorange", 0.1), linestyle=(0, (5, 10)))

        fading_blues = fading_colours(matplotlib.colormaps["Blues"], self._history_length, n_focus_history)
        previous = None
        for i, move_and_focus in enumerate(zip(move_history, focus_history)):
            colour = fading_blues[i]
            axes.plot(*move_and_focus, color=colour, zorder=i, marker=".",
                      label="focus" if i == n_focus_history - 1 else "")
            if previous is not None:
                axes.annotate("", xy=move_and_focus, xycoords="data", xytext=previous, textcoords="data",
                              arrowprops={"arrowstyle": "->", "color": colour, "shrinkA": 5, "shrinkB": 5,
                                          "connectionstyle": "arc3,rad=0.1"})
            previous = move_and_focus

        figure.legend(loc="lower right")
        figure.tight_layout()
        figure.canvas.draw()
        image = np.array(figure.canvas.buffer_rgba())[:, :, :3]
        pyplot.close(figure)
        return resize_linear_u8(image, int(frame_height * image.shape[1] / image.shape[0]), frame_height)
