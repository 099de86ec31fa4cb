"""CPU-side checks of the boundary: the shared library loads and exports every symbol the
header declares; no compute calls are made (there is no GPU in this container)."""
import ctypes as C
import os
import re

from sift_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "sift_texture.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(stx_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported():
    lib = _lib.load()
    names = _declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), n
    assert set(names) == set(_lib.EXPORTS)


def test_struct_layout_matches_header():
    assert C.sizeof(_lib.StxMetrics) == 6 * 8 + 26 * 4 + 8 + 5 * 8 + 2 * 8 + 5 * 8
    assert C.sizeof(_lib.StxParams) == 40
    assert C.sizeof(_lib.StxTiming) == 8 * 8 + 8 * 8 + 8


def test_version_and_defaults():
    lib = _lib.load()
    assert lib.stx_version().startswith(b"sift-texture")
    p = _lib.StxParams()
    lib.stx_default_params(C.byref(p))
    # the literals of /root/reference/app/passive_checker.py:103, :116
    assert (p.lbp_points, p.lbp_radius, p.canny_low, p.canny_high, p.gray_shift, p.fft_h, p.fft_w) == (
        8, 1.0, 50, 150, 14, 0, 0)


def test_score_summary_matches_numpy_mean():
    import numpy as np
    lib = _lib.load()
    rng = np.random.default_rng(1)
    for n in (0, 1, 3, 7, 8, 9, 10, 17, 129, 300):
        vals = rng.random(n)
        arr = (_lib.StxMetrics * max(n, 1))()
        for i, v in enumerate(vals):
            arr[i].texture_score = float(v)
        avg = C.c_double()
        passed = C.c_int()
        assert lib.stx_score_summary(arr, n, 0.6, C.byref(avg), C.byref(passed)) == 0
        want = float(np.mean(vals)) if n else 0.0
        assert avg.value == want
        assert passed.value == int(want >= 0.6)
