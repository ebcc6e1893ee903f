"""TEST INFRASTRUCTURE: replays what the reference's notebooks did to produce the figures they
hold as outputs (tests/golden/notebook_figures/, extracted by tests/golden/make_notebook_figures.py),
so that the same figures can be drawn from this repository's frames -- the oracle's on the CPU,
the HIP path's on the GPU -- and compared pixel by pixel.

  examples/render.ipynb       five worlds / cameras -> render.render(worlds, cameras, (300, 600))
                              -> pyplot.figure(); pyplot.imshow(frame)
  examples/environment.ipynb  DiscreteSteps(render_mode="rgb_array"): reset, render, step(8), render,
                              then cross_target(1, 11), (3, 9), (5, 7): step + render until the focus
                              plane has crossed the target; figures by show(image)

The notebooks ran matplotlib 3.8; this image has 3.10.  Figures drawn from identical frames come
out pixel-identical except for two things compare_episode_figure() sets aside: the dash phase of
axvspan's outline and the caption's move-count digit.
"""

import io
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
FIGURES = os.path.join(HERE, "golden", "notebook_figures")
INITIAL_STATE = [[5.311405, 8.66759]]  # recovered from the notebook's printed observations
FIRST_ACTION = 8                       # "print(action)" of the notebook
CROSSINGS = ((12, 1, 11), (13, 3, 9), (14, 5, 7))  # (cell, left action, right action)


def _pyplot():
    import matplotlib

    matplotlib.use("Agg")
    from matplotlib import pyplot

    return pyplot


def _inline_png(figure, pyplot):
    """What the notebook's inline backend stored: print_figure(bbox_inches="tight") at 100 dpi."""
    buffer = io.BytesIO()
    figure.savefig(buffer, format="png", dpi=100, bbox_inches="tight")
    pyplot.close(figure)
    return buffer.getvalue()


def imshow_png(frame):
    """render.ipynb cell 3: pyplot.figure(); pyplot.imshow(frame)."""
    pyplot = _pyplot()
    figure = pyplot.figure()
    pyplot.imshow(frame)
    return _inline_png(figure, pyplot)


def show_png(image):
    """environment.ipynb cell 5: show(image)."""
    pyplot = _pyplot()
    figure, axes = pyplot.subplots(figsize=(14, 14))
    axes.axis("off")
    axes.imshow(image)
    pyplot.tight_layout()
    return _inline_png(figure, pyplot)


def pixels(png):
    from matplotlib import image

    data = png if isinstance(png, bytes) else open(os.path.join(FIGURES, png), "rb").read()
    return np.round(image.imread(io.BytesIO(data))[..., :3] * 255).astype(np.int32)


def compare(ours_png, reference_name):
    """Per-pixel difference of two figures of equal size: (differing pixels, largest difference,
    total pixels)."""
    ours, reference = pixels(ours_png), pixels(reference_name)
    assert ours.shape == reference.shape, (ours.shape, reference.shape)
    difference = np.abs(ours - reference).max(axis=-1)
    return int((difference > 0).sum()), int(difference.max()), difference.size


def compare_episode_figure(ours_png, reference_name):
    """An e.render() figure against the notebook's: the 600 px frame (left 600 / 1400 of the
    image) and the plot, separately.  Two differences are known and excluded from the plot's
    count: the dash phase of axvspan's outline (a Polygon in matplotlib 3.8, a Rectangle from 3.9
    on) and the caption's digit -- the notebook was run on a revision that printed the move
    count from 1, the reference checkout (episode_visualizer.py:233) and this mirror print it from
    0.  Returns {"frame": (differing, largest), "plot": differing elsewhere, "digit_box": (h, w)}."""
    ours, reference = pixels(ours_png), pixels(reference_name)
    assert ours.shape == reference.shape, (ours.shape, reference.shape)
    difference = np.abs(ours - reference).max(axis=-1)
    split = int(ours.shape[1] * 600 / 1400)
    frame = difference[:, : split - 2]
    # the span: the pale fill ("darkorange" at alpha 0.1 over white) of the reference's figure
    fill = np.abs(reference - np.array([255, 243, 229])).max(axis=-1) <= 1
    columns = np.flatnonzero(fill.sum(axis=0) > 200)
    rows = np.flatnonzero(fill.sum(axis=1) > 20)
    x0, x1, y0, y1 = columns.min(), columns.max(), rows.min(), rows.max()
    ys, xs = np.nonzero(difference)
    plot = xs >= split - 2
    outline = (((np.abs(xs - x0) <= 6) | (np.abs(xs - x1) <= 6) | (np.abs(ys - y0) <= 6) | (np.abs(ys - y1) <= 6))
               & (xs >= x0 - 6) & (xs <= x1 + 6) & (ys >= y0 - 6) & (ys <= y1 + 6))
    rest_y, rest_x = ys[plot & ~outline], xs[plot & ~outline]
    box = (0, 0) if len(rest_y) == 0 else (int(rest_y.max() - rest_y.min() + 1), int(rest_x.max() - rest_x.min() + 1))
    return {"frame": (int((frame > 0).sum()), int(frame.max())), "plot": int(len(rest_y)), "digit_box": box}


def render_notebook_scenes():
    """(worlds, cameras, frame_shape) of render.ipynb cells 1-2."""
    from reinfocus_amd.graphics import camera, shape, shape_factory as sf, world

    targets = [2, 3.75, 5.5, 7.25, 9]
    worlds = world.Worlds(
        sf.one_rect(sf.ShapeParameters(distance=2)),
        sf.mixed(sf.ShapeParameters(distance=2), sf.ShapeParameters(distance=5.5)),
        sf.one_sphere(sf.ShapeParameters(distance=5.5)),
        sf.two_rect(sf.ShapeParameters(distance=5.5), sf.ShapeParameters(distance=9)),
        sf.one_sphere(sf.ShapeParameters(distance=9)),
    )
    frame_shape = (300, 600)
    cameras = camera.Cameras(*[
        camera.make_gpu_camera(aspect_ratio=frame_shape[1] / frame_shape[0],
                               focus_distance=float(np.interp(i, [0, 4], [10, 5])),
                               look_at=shape.v3f(0, 0, -targets[i]),
                               look_from=shape.v3f(float(np.interp(i, [0, 4], [-3, 3])), 0, 0))
        for i in range(5)])
    return worlds, cameras, frame_shape


def episode(env):
    """The notebook's episode on `env` (a single-environment DiscreteSteps with
    render_mode="rgb_array"): yields (figure name, e.render() image, observation) per render."""
    observation, _ = env.reset(state=INITIAL_STATE)
    yield "environment_cell6_0.png", env.render(), observation
    observation = env.step(FIRST_ACTION)[0]
    yield "environment_cell9_0.png", env.render(), observation
    for cell, left, right in CROSSINGS:
        high_start = env._state[0, 1] > env._state[0, 0]
        move = left if high_start else right
        index = 0
        while high_start == (env._state[0, 1] > env._state[0, 0]):
            observation = env.step(move)[0]
            yield f"environment_cell{cell}_{index}.png", env.render(), observation
            index += 1


EPISODE_FIGURES = ["environment_cell6_0.png", "environment_cell9_0.png", "environment_cell12_0.png",
                   "environment_cell12_1.png", "environment_cell13_0.png", "environment_cell14_0.png",
                   "environment_cell14_1.png", "environment_cell14_2.png"]


class OracleRenderer:
    """FastRenderer (render.py:122-257) on the CPU oracle, for the CPU-only replay: same scene
    packing classes as the product, same state growth rule, oracle render."""

    def __init__(self, oracle, samples_per_pixel=100, r_size=20, threads=8):
        from reinfocus_amd.graphics import camera, world

        self._oracle = oracle
        self._samples_per_pixel = samples_per_pixel
        self._cameras = camera.FastCameras()
        self._worlds = world.FastWorlds(r_size=r_size)
        self._ctx = type("NoContext", (), {"device": 0, "close": lambda self: None})()
        self._states = None
        self._threads = threads

    def update_targets(self, targets):
        self._worlds.update(targets)

    def update_focus_planes(self, focus_planes):
        self._cameras.update(focus_planes)

    def close(self):
        pass

    def render(self, frame_height):
        n = len(self._worlds)
        if self._states is None or len(self._states) < n * frame_height * frame_height:  # render.py:256-257
            self._states = self._oracle.seed_states(n * frame_height * frame_height, 0)
        dyn, origin, u, v, lens = self._cameras.device_data()
        return self._oracle.render(dyn, self._worlds.device_data(), frame_height, frame_height,
                                   self._samples_per_pixel, self._states,
                                   cs=self._oracle.cam_static(origin, u, v, float(lens)), n_threads=self._threads)
