"""Shape records of the general renderer (mirrors reinfocus/graphics/shape.py)."""

import dataclasses

import numpy as np

SPHERE = 0
RECTANGLE = 1


@dataclasses.dataclass
class CpuShape:
    """Parameters and 'polymorphic' type of one shape (shape.py:12-22)."""

    parameters: np.ndarray
    shape_type: int


def v2f(x=0.0, y=0.0):
    """vector.v2f (vector.py:16-26)."""
    return (np.float32(x), np.float32(y))


def v3f(x=0.0, y=0.0, z=0.0):
    """vector.v3f (vector.py:43-54)."""
    return (np.float32(x), np.float32(y), np.float32(z))


def sphere(centre, radius, texture=(16, 16)):
    """sphere.sphere (sphere.py:23-37): {x, y, z, r, fx, fy} as float32."""
    return CpuShape(np.array([*centre, radius, *texture], dtype=np.float32), SPHERE)


def rectangle(x_span, y_span, z_pos, texture=(16, 16)):
    """rectangle.rectangle (rectangle.py:26-46): {x_min, x_max, y_min, y_max, z, fx, fy}."""
    return CpuShape(np.array([*x_span, *y_span, z_pos, *texture], dtype=np.float32), RECTANGLE)
