"""The scenes of the reference's notebooks (mirrors reinfocus/graphics/shape_factory.py)."""

import math
from typing import NamedTuple

from reinfocus_amd.graphics import shape


class ShapeParameters(NamedTuple):
    """shape_factory.py:14-26."""

    distance: float = 10.0
    size: float = 0.0
    r_size: float = 20.0
    texture_f: tuple = (16, 16)


def get_absolute_size(parameters):
    """shape_factory.py:29-41: explicit size, else distance * tan(r_size / 2)."""
    if parameters.size != 0.0:
        return parameters.size
    return parameters.distance * math.tan(math.radians(parameters.r_size / 2))


_OFFSET = math.tan(math.radians(15))  # shapes sit 15 degrees off the optical axis


def _sphere_at(x, parameters):
    return shape.sphere(shape.v3f(x, 0, -parameters.distance), get_absolute_size(parameters),
                        shape.v2f(*parameters.texture_f))


def _rect_at(x, parameters):
    size = get_absolute_size(parameters)
    return shape.rectangle(shape.v2f(x - size, x + size), shape.v2f(-size, size), -parameters.distance,
                           shape.v2f(*parameters.texture_f))


def one_sphere(parameters=ShapeParameters()):
    """shape_factory.py:44-62."""
    return [_sphere_at(0, parameters)]


def two_sphere(left_parameters=ShapeParameters(20.0), right_parameters=ShapeParameters(5.0)):
    """shape_factory.py:65-101."""
    return [_sphere_at(-left_parameters.distance * _OFFSET, left_parameters),
            _sphere_at(right_parameters.distance * _OFFSET, right_parameters)]


def one_rect(parameters=ShapeParameters()):
    """shape_factory.py:104-123."""
    return [_rect_at(0, parameters)]


def two_rect(left_parameters=ShapeParameters(20.0), right_parameters=ShapeParameters(5.0)):
    """shape_factory.py:126-163."""
    return [_rect_at(-(left_parameters.distance * _OFFSET), left_parameters),
            _rect_at(right_parameters.distance * _OFFSET, right_parameters)]


def mixed(left_parameters=ShapeParameters(5.0), right_parameters=ShapeParameters()):
    """shape_factory.py:166-196: a sphere on the left, a rectangle on the right."""
    return [_sphere_at(-left_parameters.distance * _OFFSET, left_parameters),
            _rect_at(right_parameters.distance * _OFFSET, right_parameters)]
