"""FastCameras: per-environment thin-lens camera parameters.

Mirrors reinfocus/graphics/camera.py:94-179.  The reference packs one environment at
a time with ~10 numpy calls each; here the same arithmetic (numpy-1.26 promotion:
`python_float * float32 -> float64`, rounded to float32 once) is vectorised over
environments.  Output layout is camera.py:39-56: (dynamic float32[n,3,3] =
[lower_left, horizontal, vertical], origin, u, v, lens_radius float64).
"""

import math

import numpy as np

from reinfocus_amd.graphics import device_data

f32 = np.float32


def _v3(x):
    return np.asarray(x, dtype=np.float32).reshape(3)


def _norm(v):
    # vector.norm_v3f (vector.py:342-351): v * (1.0 / float(||v||_f32)), the Python float
    # is demoted to float32 by the array multiply.
    length = float(np.linalg.norm(v))
    return v * f32(1.0 / length)


class FastCameras(device_data.DeviceData):
    """All cameras share everything but the focus distance (camera.py:94-130)."""

    def __init__(self, aspect_ratio=1, look_from=(0, 0, 0), look_at=(0, 0, -10), up=(0, 1, 0),
                 aperture=0.1, vfov=30):
        super().__init__()
        self._look_from = _v3(look_from)
        self._half_aperture = np.divide(aperture, 2.0)  # numpy.float64, as the reference
        self._half_height = math.tan((vfov * math.pi / 180.0) / 2.0)
        self._half_width = aspect_ratio * self._half_height
        self._w = _norm(self._look_from - _v3(look_at))
        self._u = _norm(np.cross(_v3(up), self._w).astype(np.float32))
        self._v = np.cross(self._w, self._u).astype(np.float32)

    def _make_device_data(self, data):
        """camera.py:132-179 for every focus plane at once."""
        fp = np.asarray(data, dtype=np.float32)
        fp64 = fp.astype(np.float64)
        # smul_v3f(u, half_width * focus_plane): f64 product -> f32 scalar -> f32 multiply
        a = (self._half_width * fp64).astype(np.float32)[:, None] * self._u[None, :]
        b = (self._half_height * fp64).astype(np.float32)[:, None] * self._v[None, :]
        c = fp[:, None] * self._w[None, :]
        # add_v3f: numpy.sum over the three vectors, left to right in f32
        lower_left = self._look_from[None, :] - ((a + b) + c)
        horizontal = ((2.0 * self._half_width) * fp64).astype(np.float32)[:, None] * self._u[None, :]
        vertical = ((2.0 * self._half_height) * fp64).astype(np.float32)[:, None] * self._v[None, :]
        dyn = np.stack([lower_left, horizontal, vertical], axis=1).astype(np.float32)
        return (dyn, self._look_from, self._u, self._v, self._half_aperture)


def make_gpu_camera(aperture=0.1, aspect_ratio=1, focus_distance=10, look_at=(0, 0, -10), look_from=(0, 0, 0),
                    up=(0, 1, 0), vfov=30):
    """camera.make_gpu_camera (camera.py:182-226): the 7-tuple (lower_left, horizontal,
    vertical, origin, u, v, lens radius) of one camera; float32 vectors, float64 radius."""
    look_from = _v3(look_from)
    half_height = math.tan((vfov * math.pi / 180.0) / 2.0)
    half_width = aspect_ratio * half_height
    w = _norm(look_from - _v3(look_at))
    u = _norm(np.cross(_v3(up), w).astype(np.float32))
    v = np.cross(w, u).astype(np.float32)
    # smul_v3f(u, half_width * focus_distance): the Python-float product is demoted to f32
    # by the array multiply; add_v3f sums the three vectors left to right in f32
    a = f32(half_width * focus_distance) * u
    b = f32(half_height * focus_distance) * v
    c = f32(focus_distance) * w
    return (look_from - ((a + b) + c), f32(2.0 * half_width * focus_distance) * u,
            f32(2.0 * half_height * focus_distance) * v, look_from, u, v, np.divide(aperture, 2.0))


class Cameras:
    """camera.Cameras (camera.py:59-91): per-environment cameras packed as the float64[n, 19]
    rows numpy.hstack produces from the float32 vectors and the float64 lens radius."""

    def __init__(self, *cameras):
        self._d_cameras = np.hstack(
            [np.asarray([cam[k] for cam in cameras], dtype=np.float64) for k in range(6)]
            + [np.reshape([cam[6] for cam in cameras], (len(cameras), 1)).astype(np.float64)]
        )

    def __len__(self):
        return self._d_cameras.shape[0]

    def device_data(self):
        return self._d_cameras
