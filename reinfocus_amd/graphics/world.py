"""FastWorlds: one z-aligned checkerboard square per environment.

Mirrors reinfocus/graphics/world.py:85-123 with shape_factory.get_absolute_size
(shape_factory.py:29-41): half_side = float32(float64(target) * tan(radians(r_size/2))),
z = -target; packed as float32[n, 2] (rectangle.py:21-23 FH_RADIUS, FH_ZPOS).
"""

import math

import numpy as np

from reinfocus_amd.graphics import device_data


class FastWorlds(device_data.DeviceData):
    """Targets that always subtend r_size degrees of field of view (world.py:90-98)."""

    def __init__(self, r_size: float = 20):
        super().__init__()
        self._r_size = r_size

    def _make_device_data(self, data):
        targets = np.asarray(data, dtype=np.float32)
        half = (targets.astype(np.float64) * math.tan(math.radians(self._r_size / 2))).astype(np.float32)
        return np.ascontiguousarray(np.stack([half, -targets], axis=1), dtype=np.float32)


class Worlds:
    """world.Worlds (world.py:27-82): per-environment shape lists packed as
    (parameters float32[n, most, width], types int32[n, most], sizes int32[n])."""

    def __init__(self, *env_shapes):
        sizes = np.array([len(shapes) for shapes in env_shapes], dtype=np.int32)
        self._num_envs = len(env_shapes)
        most = int(max(sizes))
        width = max(max(len(s.parameters) for s in shapes) for shapes in env_shapes)
        parameters = np.zeros((self._num_envs, most, width), dtype=np.float32)
        types = np.zeros((self._num_envs, most), dtype=np.int32)
        for e, shapes in enumerate(env_shapes):
            for i, s in enumerate(shapes):
                parameters[e, i, : len(s.parameters)] = s.parameters
                types[e, i] = s.shape_type
        self._data = (parameters, types, sizes)

    def __len__(self):
        return self._num_envs

    def device_data(self):
        return self._data
