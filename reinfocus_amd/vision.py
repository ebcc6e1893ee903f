"""Focus measure on MI355X.

Mirrors reinfocus/vision.py:11-39 (same names and argument meaning).  The OpenCV
chain cvtColor(RGB2GRAY) -> medianBlur(3) -> Laplacian(CV_8U) -> .var() runs as one
HIP kernel per batch (rf_focus); frames that are still resident on the GPU
(graphics.render.DeviceFrames) are scored in place, host arrays are uploaded first.
"""

import numpy as np

from reinfocus_amd import _native
from reinfocus_amd.graphics import render

GRAY_MODE = _native.GRAY_15BIT  # opencv-python ~=4.9 (pyproject.toml:33); 14 = OpenCV 2/3

_scratch = None


def _scratch_context():
    """Context that scores host arrays (one per process, created on first use, released by
    release() or at interpreter exit)."""
    global _scratch
    if _scratch is None:
        import atexit

        _scratch = _native.Context()
        atexit.register(release)
    return _scratch


def release():
    """Frees the scratch context's device memory; the next host-array call makes a new one."""
    global _scratch
    if _scratch is not None:
        _scratch.close()
        _scratch = None


def focus_values(images):
    """vision.py:28-39: one focus value per RGB image, as a list of floats."""
    if isinstance(images, render.DeviceFrames) and images.is_resident():
        n, h, w = images.shape[:3]
        return list(images.device_context().focus(n, h, w, GRAY_MODE))
    host = np.ascontiguousarray(np.asarray(images), dtype=np.uint8)
    assert host.ndim == 4 and host.shape[3] == 3, "images must be uint8[n, h, w, 3]"
    ctx = _scratch_context()
    ctx.upload_frames(host)
    return list(ctx.focus(host.shape[0], host.shape[1], host.shape[2], GRAY_MODE))


def focus_value(image):
    """vision.py:11-25: focus value of one RGB image."""
    image = np.asarray(image)
    return float(focus_values(image[None, ...])[0])
