"""The figures the reference's notebooks hold as outputs -- 13 images the real reference rendered
(numba on CUDA, OpenCV 4.9) -- redrawn from the ORACLE's frames and compared pixel by pixel
(tests/notebook_figures.py).  This pins the oracle, fast path and general renderer alike, to the
reference's own pixels: 8 FastRenderer frames of 600 x 600 x 100 spp along a 7-step episode
(all of them identical to within 2 pixels of 358 737, by one grey level) and 5 general-renderer
frames of 300 x 600 x 100 spp (identical except for 4 - 87 figure pixels of 163 392 by at most
3 levels: about one sample in a million lands on the other side of a checker or shape edge --
the reference's arithmetic on CUDA contracts a * b + c and uses CUDA's libm).  The plots of the
episode figures (focus positions and focus values of every step, captions, arrows) are
identical too, apart from the dash phase of axvspan's outline (matplotlib 3.8 vs 3.10) and the
caption's move-count digit (the notebook's revision counted from 1).
tests/test_gpu_notebook_figures.py does the same with the HIP path's frames."""

import numpy as np

from tests import notebook_figures as nf


def test_general_renderer_reproduces_render_notebook_figures(oracle):
    worlds, cameras, (h, w) = nf.render_notebook_scenes()
    params, types, sizes = worlds.device_data()
    params = np.ascontiguousarray(np.pad(params, ((0, 0), (0, 0), (0, max(0, 7 - params.shape[2])))))
    states = oracle.seed_states(5 * h * w, 0)  # render.py:115: fresh seed-0 states per call
    frames = oracle.render_general(cameras.device_data(), params, types, sizes, h, w, 100, states, n_threads=8)
    worst = 0
    for i, frame in enumerate(frames):
        differing, largest, total = nf.compare(nf.imshow_png(frame), f"render_cell3_{i}.png")
        assert total == 296 * 552
        assert differing <= 120 and largest <= 3, (i, differing, largest)
        worst = max(worst, differing)
    assert worst > 0  # not bit-identical, and we say so: see the module docstring


def test_fast_path_reproduces_environment_notebook_figures(oracle, monkeypatch):
    from reinfocus_amd import vision
    from reinfocus_amd.environments import harness, state_observer
    from reinfocus_amd.graphics import render

    monkeypatch.setattr(render, "FastRenderer",
                        lambda samples_per_pixel=100, r_size=20, **_: nf.OracleRenderer(oracle, samples_per_pixel, r_size))
    monkeypatch.setattr(vision, "focus_values", lambda frames: list(oracle.focus_values(np.asarray(frames), 15, 8)))
    state_observer._focus_extrema.cache_clear()
    try:
        env = harness.DiscreteSteps(render_mode="rgb_array")
        names = []
        for name, image, observation in nf.episode(env):
            names.append(name)
            assert image.shape == (600, 1400, 3) and observation.shape == (4,)
            found = nf.compare_episode_figure(nf.show_png(image), name)
            # the 600 px frame: identical to within a few pixels by one level
            assert found["frame"][0] <= 6 and found["frame"][1] <= 1, (name, found)
            # the plot (positions, focus values, arrows, labels): nothing differs but one digit
            assert found["plot"] <= 250 and found["digit_box"][0] <= 20 and found["digit_box"][1] <= 16, (name, found)
        assert names == nf.EPISODE_FIGURES
    finally:
        state_observer._focus_extrema.cache_clear()
