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
    (episode_visualizer.py:84-301; same constructor arguments).

    The plots are built on matplotlib's object API with an Agg canvas of their own (no pyplot
    state, nothing to close), one helper per plot element."""

    FRAME = 600  # pixels: height of a row and side of the rendering (episode_visualizer.py:197, :206)

    def __init__(self, num_envs, target_index, focus_plane_index, focus_value_index, renderer, limits,
                 ender=None, history_length=10, target_radius=None):
        self._renderer = renderer
        self._ender = ender
        self._limits = limits
        self._target_radius = target_radius
        self._history_length = history_length
        self._num_envs = num_envs
        self._columns = {"target": target_index, "plane": focus_plane_index, "value": focus_value_index}
        self._current_moves = np.zeros(num_envs, dtype=np.int32)
        self._targets = np.zeros(num_envs, dtype=np.float32)
        self._move_histories = histories.Histories(num_envs, history_length)
        self._focus_histories = histories.Histories(num_envs, history_length)

    # -- bookkeeping ------------------------------------------------------------------------------
    def _selection(self, indices):
        return np.full(self._num_envs, True) if indices is None else indices

    def _record(self, states, observations, selected):
        self._move_histories.append_events(states[:, self._columns["plane"]], selected)
        self._focus_histories.append_events(observations[:, self._columns["value"]], selected)

    def step(self, states, observations, indices=None):
        """One timestep of the selected environments (episode_visualizer.py:133-155)."""
        selected = self._selection(indices)
        self._current_moves[selected] += 1
        self._record(states, observations, selected)

    def reset(self, states, observations, indices=None):
        """The selected environments started new episodes (episode_visualizer.py:157-185).  The
        reference appends the first focus value to every history here (:185), which only works
        while all environments reset together; this appends to the selected ones."""
        selected = self._selection(indices)
        self._current_moves[selected] = 0
        self._targets[selected] = states[:, self._columns["target"]]
        self._move_histories.reset(selected)
        self._focus_histories.reset(selected)
        self._record(states, observations, selected)

    # -- drawing ----------------------------------------------------------------------------------
    def visualize(self):
        """uint8[num_envs * 600, 600 + plot width, 3] (episode_visualizer.py:188-201)."""
        frames = np.asarray(self._renderer.render(self.FRAME))
        # zip(renderings, graphs) in the reference: after a partial auto-reset render the shared
        # renderer holds only the environments that were reset, and only that many rows come out
        rows = [np.concatenate([frames[i], self._visualize_single_history(i)], axis=1)
                for i in range(min(len(frames), self._num_envs))]
        return np.concatenate(rows, axis=0)

    def _caption(self, env_index):
        caption = f"focus position {self._current_moves[env_index]}\n"
        return caption + (self._ender.status(env_index) if self._ender is not None else "")

    def _draw_target(self, axes, target):
        axes.axvline(x=target, linestyle=":", color="darkorange", label="target")
        radius = self._target_radius
        if radius is not None and radius > 0.0:
            axes.axvspan(target - radius, target + radius, edgecolor="darkorange", facecolor=("darkorange", 0.1),
                         linestyle=(0, (5, 10)))

    def _draw_trail(self, axes, moves, values):
        """The remembered (position, value) points, oldest palest, joined by curved arrows."""
        import matplotlib

        count = len(values)
        shades = fading_colours(matplotlib.colormaps["Blues"], self._history_length, count)
        points = list(zip(moves, values))
        for order, (point, shade) in enumerate(zip(points, shades)):
            axes.plot(*point, color=shade, zorder=order, marker=".", label="focus" if order == count - 1 else "")
            if order:
                axes.annotate("", xy=point, xytext=points[order - 1], xycoords="data", textcoords="data",
                              arrowprops=dict(arrowstyle="->", color=shade, shrinkA=5, shrinkB=5,
                                              connectionstyle="arc3,rad=0.1"))

    def _visualize_single_history(self, env_index, frame_height=None):
        """The performance plot of one environment (episode_visualizer.py:203-301), scaled to the
        height of a row."""
        from matplotlib.backends.backend_agg import FigureCanvasAgg
        from matplotlib.figure import Figure

        frame_height = self.FRAME if frame_height is None else frame_height
        figure = Figure()  # rcParams size and dpi, as pyplot.subplots() in the reference
        canvas = FigureCanvasAgg(figure)
        axes = figure.add_subplot()
        axes.set_xlim(*self._limits)
        axes.set_ylim(-1.0, 1.0)
        axes.set_xlabel(self._caption(env_index))
        axes.set_ylabel("focus value")
        self._draw_target(axes, self._targets[env_index])
        self._draw_trail(axes, self._move_histories.get_history(env_index),
                         self._focus_histories.get_history(env_index))
        figure.legend(loc="lower right")
        figure.tight_layout()
        canvas.draw()
        plot = np.asarray(canvas.buffer_rgba())[:, :, :3]
        return resize_linear_u8(plot, int(frame_height * plot.shape[1] / plot.shape[0]), frame_height)
