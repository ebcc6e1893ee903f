"""FastRenderer on MI355X.

Mirrors reinfocus/graphics/render.py:122-257: same constructor, update_targets,
update_focus_planes and render(frame_height).  The numba kernel launch + full-frame
copy_to_host of the reference (render.py:178-188) becomes one rf_render call that
leaves the frames in HBM; render() returns a DeviceFrames handle that behaves like
the uint8[N, h, h, 3] array (numpy.asarray / len / iteration / indexing materialise it)
and that vision.focus_values scores on the device without the 3 B/pixel D2H.
"""

import weakref

import numpy as np

from reinfocus_amd import _native
from reinfocus_amd.graphics import camera
from reinfocus_amd.graphics import cutil
from reinfocus_amd.graphics import world


class DeviceFrames:
    """uint8[N, h, w, 3] frames of one render, resident in the renderer's HBM buffer."""

    def __init__(self, renderer, shape, generation):
        self._renderer = renderer
        self.shape = tuple(shape)
        self.dtype = np.dtype(np.uint8)
        self.ndim = 4
        self._generation = generation
        self._host = None

    # -- device side ----------------------------------------------------------------------
    def is_resident(self):
        """True while the renderer's frame buffer still holds this render."""
        return self._renderer is not None and self._renderer._generation == self._generation

    def device_context(self):
        assert self.is_resident()
        return self._renderer._ctx

    # -- host side ------------------------------------------------------------------------
    def numpy(self):
        """Copies the frames to the host (once)."""
        if self._host is None:
            assert self.is_resident(), "frames were overwritten by a later render"
            self._host = self._renderer._ctx.get_frames(self.shape)
        return self._host

    def _detach(self):
        """Called by the renderer before it overwrites its frame buffer."""
        if self._host is None and self.is_resident():
            self.numpy()
        self._renderer = None

    def __array__(self, dtype=None, copy=None):
        a = self.numpy()
        return a if dtype is None else a.astype(dtype)

    def __len__(self):
        return self.shape[0]

    def __iter__(self):
        return iter(self.numpy())

    def __getitem__(self, item):
        return self.numpy()[item]

    def astype(self, dtype):
        return self.numpy().astype(dtype)


class FastRenderer:
    """Produces images of focus scenes (render.py:122-145)."""

    def __init__(self, block_shape=(1, 16, 16), samples_per_pixel=100, r_size=20, device=None,
                 first_state_index=0, host_frames=False):
        """block_shape is accepted for signature parity; the gfx950 kernel fixes its own
        launch geometry (256-thread blocks, lanes along x).  `device` / `first_state_index`
        are extensions for one-process-per-GPU sharding (default: LOCAL_RANK, 0).
        host_frames=True is the literal drop-in (INTEGRATION.md's stub): render() returns the
        numpy array itself -- rf_render with host_out, one device-to-host copy of every frame per
        call, as the reference's copy_to_host (render.py:188) -- instead of a DeviceFrames handle."""
        self._block_shape = cutil.check_block_shape(block_shape)
        self._samples_per_pixel = samples_per_pixel
        self._cameras = camera.FastCameras()
        self._worlds = world.FastWorlds(r_size=r_size)
        self._ctx = _native.Context(device)
        self._first_state_index = int(first_state_index)
        self._host_frames = bool(host_frames)
        self._n_states = 0  # render.py:145 _random_states = None
        self._uploaded = (-1, -1)
        self._generation = 0
        self._last_frames = None

    def close(self):
        """Releases the rf_ctx (device memory); frames still referenced are copied out first."""
        last = self._last_frames() if self._last_frames is not None else None
        if last is not None:
            last._detach()
        self._ctx.close()

    def update_targets(self, targets):
        """render.py:147-154."""
        self._worlds.update(targets)

    def update_focus_planes(self, focus_planes):
        """render.py:156-163."""
        self._cameras.update(focus_planes)

    def _upload_scene(self):
        version = (self._worlds.version, self._cameras.version)
        rect = self._worlds.device_data()  # AssertionError before any update, as the reference
        dyn, origin, u, v, lens_radius = self._cameras.device_data()
        if version != self._uploaded:
            assert len(dyn) == len(rect), "targets and focus planes differ in length"
            self._ctx.set_scene(dyn, rect, origin, u, v, lens_radius)
            self._uploaded = version

    def _make_random_states(self, grid_shape):
        """render.py:248-257: (re)seed from seed 0 only when more states are needed."""
        total = int(np.prod(grid_shape))
        if self._n_states < total:
            self._ctx.seed(total, 0, self._first_state_index)
            self._n_states = total

    def render(self, frame_height: int):
        """render.py:165-188; returns DeviceFrames of shape (N, h, h, 3)."""
        n = len(self._worlds)
        grid_shape = (n, frame_height, frame_height)
        self._upload_scene()
        self._make_random_states(grid_shape)
        last = self._last_frames() if self._last_frames is not None else None
        if last is not None:
            last._detach()
        if self._host_frames:
            self._generation += 1
            return self._ctx.render(n, frame_height, frame_height, self._samples_per_pixel, to_host=True)
        self._ctx.render(n, frame_height, frame_height, self._samples_per_pixel)
        self._generation += 1
        frames = DeviceFrames(self, grid_shape + (3,), self._generation)
        self._last_frames = weakref.ref(frames)
        return frames


def render(world_data, cameras, frame_shape=(300, 600), block_shape=(1, 16, 16), samples_per_pixel=100,
           device=None):
    """render.render (render.py:88-119): ray traced images uint8[N, H, W, 3] of world_data
    (world.Worlds) seen by cameras (camera.Cameras); fresh seed-0 RNG states per call, up
    to 50 bounces per sample.  block_shape is checked as the reference's launcher would
    (cutil.check_block_shape); the gfx950 kernel fixes its own launch geometry."""
    cutil.check_block_shape(block_shape)
    parameters, types, sizes = world_data.device_data()
    ctx = _native.Context(device)
    try:
        return ctx.render_general(cameras.device_data(), parameters, types, sizes, frame_shape[0], frame_shape[1],
                                  samples_per_pixel)
    finally:
        ctx.close()
