"""Update-only-when-changed holder for per-environment scene inputs.

Mirrors reinfocus/graphics/device_data.py:16-78 (same method names, same caching and
error behaviour).  The transformed data here is the packed float32 host array; the
upload itself happens in FastRenderer, which owns the device context.
"""

import abc

import numpy as np


class DeviceData(abc.ABC):
    """Transforms some per-environment inputs, but only when they change."""

    def __init__(self):
        self._data = None
        self._d_device_data = None
        self.version = 0  # bumped on every re-pack; FastRenderer uploads on change

    def __len__(self) -> int:
        """Length of the last set of inputs, 0 before any update (device_data.py:27-35)."""
        return len(self._data) if self._data is not None else 0

    def device_data(self):
        """The packed data; AssertionError before the first update (device_data.py:37-45)."""
        assert self._d_device_data is not None
        return self._d_device_data

    def update(self, data):
        """Re-packs only if `data` differs from the last update (device_data.py:47-66).

        Same comparison as the reference (shape, then element-wise ==, so a NaN always
        re-packs), vectorised instead of a Python-level all()."""
        data = np.asarray(data, dtype=np.float32)
        if (
            self._data is not None
            and self._data.shape == data.shape
            and bool(np.all(self._data == data))
        ):
            return
        self._data = data.copy()
        self._d_device_data = self._make_device_data(self._data)
        self.version += 1

    @abc.abstractmethod
    def _make_device_data(self, data):
        """The expensive transform (device_data.py:68-78)."""
