"""The oracle's restatement of the OpenCV chain (vision.py:23-25) against independent
implementations: scipy.ndimage for median / Laplacian, numpy for var, closed forms for
gray.  OpenCV itself is not installed in this image."""

import numpy as np
import pytest
from scipy import ndimage


@pytest.fixture(scope="module")
def images():
    rng = np.random.default_rng(11)
    out = [rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
           for h, w in ((1, 1), (1, 7), (7, 1), (2, 2), (3, 5), (16, 16), (33, 35), (64, 48))]
    smooth = np.zeros((40, 40, 3), dtype=np.uint8)
    yy, xx = np.mgrid[0:40, 0:40]
    smooth[..., 0] = (xx * 6) % 256
    smooth[..., 1] = (yy * 6) % 256
    smooth[..., 2] = ((xx + yy) * 3) % 256
    return out + [smooth]


def test_gray_coefficients(oracle, images):
    for img in images:
        r, g, b = (img[..., k].astype(np.int64) for k in range(3))
        want15 = ((r * 9798 + g * 19235 + b * 3735 + (1 << 14)) >> 15).astype(np.uint8)
        want14 = ((r * 4899 + g * 9617 + b * 1868 + (1 << 13)) >> 14).astype(np.uint8)
        assert np.array_equal(oracle.gray(img, 15), want15)
        assert np.array_equal(oracle.gray(img, 14), want14)
    # white stays white, primaries are the rounded luma weights
    px = np.array([[[255, 255, 255], [255, 0, 0], [0, 255, 0], [0, 0, 255]]], dtype=np.uint8)
    assert oracle.gray(px, 15).tolist() == [[255, 76, 150, 29]]


def test_median_matches_scipy(oracle, images):
    for img in images:
        gray = oracle.gray(img)
        want = ndimage.median_filter(gray, size=3, mode="nearest")  # BORDER_REPLICATE
        assert np.array_equal(oracle.median3(gray), want)


def test_laplacian_matches_scipy(oracle, images):
    kernel = np.array([[0, 1, 0], [1, -4, 1], [0, 1, 0]])
    for img in images:
        gray = oracle.gray(img)
        want = ndimage.correlate(gray.astype(np.int32), kernel, mode="mirror")  # REFLECT_101
        assert np.array_equal(oracle.laplacian_u8(gray), np.clip(want, 0, 255).astype(np.uint8))


def test_var_matches_numpy_bit_for_bit(oracle):
    rng = np.random.default_rng(3)
    for shape in ((1, 1), (3, 5), (10, 10), (127, 129), (300, 300), (600, 600)):
        img = rng.integers(0, 256, size=shape, dtype=np.uint8)
        assert oracle.var_u8(img) == img.var()


def test_focus_value_chain(oracle, images):
    kernel = np.array([[0, 1, 0], [1, -4, 1], [0, 1, 0]])
    for img in images:
        gray = oracle.gray(img)
        med = ndimage.median_filter(gray, size=3, mode="nearest")
        lap = np.clip(ndimage.correlate(med.astype(np.int32), kernel, mode="mirror"), 0, 255).astype(np.uint8)
        assert oracle.focus_value(img) == lap.var()
    batch = np.stack([images[-1], images[-1][::-1].copy()])
    fv = oracle.focus_values(batch)
    assert fv[0] == oracle.focus_value(batch[0]) and fv[1] == oracle.focus_value(batch[1])
