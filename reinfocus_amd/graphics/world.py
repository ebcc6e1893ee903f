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
