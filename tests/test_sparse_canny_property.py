"""The claim the GPU's sparse placement of Canny's non-maximum suppression rests on, checked on the CPU with the numpy
restatement of cv2.Canny (/root/reference/app/passive_checker.py:116): every edge pixel lies in the set a flood of the
RAW double-threshold maps reaches (magnitude above low, seeded by magnitude above high), so suppressing only inside
that set and flooding again gives the same edge map as suppressing everywhere."""
import numpy as np
import pytest
from scipy import ndimage

from cases import SEED
from oracle.texture_oracle import np_canny, np_gray
from sift_amd.synth import KINDS, synth_frame

EIGHT = np.ones((3, 3), dtype=int)


def _flood(seed, allowed):
    lab, n = ndimage.label(allowed, structure=EIGHT)
    if not n:
        return np.zeros_like(allowed)
    hit = ndimage.maximum(seed & allowed, lab, index=np.arange(1, n + 1)) > 0
    return np.concatenate([[False], hit])[lab]


def _sparse_canny(gray, low, high):
    g = np.pad(gray.astype(np.int64), 1, mode="edge")
    gx = (g[:-2, 2:] - g[:-2, :-2]) + 2 * (g[1:-1, 2:] - g[1:-1, :-2]) + (g[2:, 2:] - g[2:, :-2])
    gy = (g[2:, :-2] - g[:-2, :-2]) + 2 * (g[2:, 1:-1] - g[:-2, 1:-1]) + (g[2:, 2:] - g[:-2, 2:])
    mag = np.abs(gx) + np.abs(gy)
    reached = _flood(mag > high, mag > low)          # the superset
    _, nms = np_canny(gray, low, high)               # the suppression's verdict per pixel (0 / candidate / strong)
    cand2 = (nms > 0) & reached                      # ... consulted inside the superset only
    strong2 = (nms == 2) & reached
    return np.where(_flood(strong2, cand2), 255, 0).astype(np.uint8), reached


@pytest.mark.parametrize("kind", KINDS)
def test_suppression_inside_the_flooded_set_gives_the_same_edges(kind):
    for h, w in ((1, 1), (3, 7), (64, 64), (97, 131), (240, 320)):
        gray = np_gray(synth_frame(kind, SEED, 21, 0, h, w))
        for low, high in ((50, 150), (0, 0), (20, 60), (120, 400)):
            want, _ = np_canny(gray, low, high)
            got, reached = _sparse_canny(gray, low, high)
            assert np.array_equal(got, want), (kind, h, w, low, high)
            assert not (want.astype(bool) & ~reached).any()


def test_random_images_and_long_contours():
    rng = np.random.default_rng(3)
    for _ in range(6):
        h, w = int(rng.integers(8, 90)), int(rng.integers(8, 120))
        gray = rng.integers(0, 256, (h, w), dtype=np.uint8)
        if rng.random() < 0.5:  # smooth it: sparse candidates, real contours
            gray = ndimage.uniform_filter(gray, 5)
        want, _ = np_canny(gray, 50, 150)
        got, _ = _sparse_canny(gray, 50, 150)
        assert np.array_equal(got, want)
    g = np.full((120, 160), 40, np.uint8)
    for y in range(4, 116, 6):
        g[y, 4:156] = 58
    g[4:116, 4] = 58
    g[10:14, 10:14] = 255
    want, _ = np_canny(g, 50, 150)
    got, _ = _sparse_canny(g, 50, 150)
    assert np.array_equal(got, want) and want.any()
