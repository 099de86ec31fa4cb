"""Opportunistic parity against the libraries the reference calls (app/passive_checker.py:100-123 uses cv2,
scikit-image and numpy; requirements.txt pins no versions).  Neither cv2 nor scikit-image is importable where this
repository is built, so these tests skip there; on a machine that has them they turn the "parity unpinned" stages
(gray, high-pass, Gaussian, Canny, Laplacian, resize) into pinned ones, for the CPU oracle and for the HIP path.
They use this repository's own harness only."""
import numpy as np
import pytest

from cases import SEED
from oracle import texture_oracle as O
from sift_amd.synth import KINDS, synth_frame

SHAPES = [(1, 1), (5, 7), (64, 64), (45, 77), (130, 257), (240, 320)]
HP_KERNEL = np.array([[-1, -1, -1], [-1, 8, -1], [-1, -1, -1]])


def _frames():
    for kind in KINDS:
        for i, (h, w) in enumerate(SHAPES):
            yield kind, synth_frame(kind, SEED + 9, 60, i, h, w)


def _cv2_gray_shift(cv2):
    """Which fixed-point width this OpenCV build's BGR2GRAY uses: 15 (4.x) or 14 (3.x)."""
    px = np.array([[[176, 64, 33]]], np.uint8)  # 67 with the 15-bit weights, 68 with the 14-bit ones
    v = int(cv2.cvtColor(px, cv2.COLOR_BGR2GRAY)[0, 0])
    assert v in (67, 68), v
    return 15 if v == 67 else 14


def test_oracle_vs_opencv(coracle):
    cv2 = pytest.importorskip("cv2")
    shift = _cv2_gray_shift(cv2)
    print("OpenCV", cv2.__version__, "BGR2GRAY fixed-point width:", shift)
    assert shift == coracle.default_params().gray_shift, "the library's default gray_shift does not match this OpenCV"
    for kind, f in _frames():
        gray = cv2.cvtColor(f, cv2.COLOR_BGR2GRAY)
        assert np.array_equal(coracle.gray(f, shift), gray), ("gray", kind, f.shape)
        assert np.array_equal(coracle.highpass(gray), cv2.filter2D(gray, -1, HP_KERNEL)), ("filter2D", kind, f.shape)
        assert np.array_equal(coracle.gauss5(gray), cv2.GaussianBlur(gray, (5, 5), 0)), ("GaussianBlur", kind, f.shape)
        assert np.array_equal(coracle.canny(gray)[0], cv2.Canny(gray, 50, 150)), ("Canny", kind, f.shape)
        lap = cv2.Laplacian(gray, cv2.CV_64F)
        s1, s2, _ = coracle.quality_sums(gray)
        assert (s1, s2) == (int(lap.sum()), int((lap * lap).sum())), ("Laplacian", kind, f.shape)
        if min(f.shape[:2]) >= 2:
            rgb = cv2.cvtColor(f, cv2.COLOR_BGR2RGB)
            want = cv2.resize(rgb, (80, 80))
            assert np.array_equal(O.np_resize_linear_u8(rgb, 80, 80), want), ("resize", kind, f.shape)


def test_oracle_vs_skimage(coracle):
    feature = pytest.importorskip("skimage.feature")
    for kind, f in _frames():
        gray = coracle.gray(f)
        for P, R in ((8, 1.0), (16, 2.0), (24, 3.0)):
            want = feature.local_binary_pattern(gray, P, R, method="uniform")
            assert np.array_equal(coracle.lbp(gray, P, R)[0], want.astype(np.uint8)), (kind, f.shape, P)


def test_oracle_metrics_vs_reference_calls(coracle):
    """The five metrics exactly as the reference computes them (the calls of passive_checker.py:100-123)."""
    cv2 = pytest.importorskip("cv2")
    feature = pytest.importorskip("skimage.feature")
    p = coracle.default_params()
    p.gray_shift = _cv2_gray_shift(cv2)
    for kind, f in _frames():
        gray = cv2.cvtColor(f, cv2.COLOR_BGR2GRAY)
        want = {
            "lbp_variance": float(np.var(feature.local_binary_pattern(gray, 8, 1, method="uniform"))),
            "noise_level": float(np.std(cv2.filter2D(gray, -1, HP_KERNEL))),
            "compression_artifacts": float(np.mean(cv2.absdiff(gray, cv2.GaussianBlur(gray, (5, 5), 0)))),
            "edge_density": float(np.sum(cv2.Canny(gray, 50, 150) > 0) / gray.size),
            "frequency_anomaly": float(np.std(np.log(np.abs(np.fft.fftshift(np.fft.fft2(gray))) + 1))),
        }
        m = coracle.analyze_frame(f, p)
        for k, v in want.items():
            assert getattr(m, k) == pytest.approx(v, rel=1e-9, abs=1e-12), (k, kind, f.shape)


@pytest.mark.gpu
def test_hip_planes_vs_opencv_and_skimage():
    cv2 = pytest.importorskip("cv2")
    feature = pytest.importorskip("skimage.feature")
    from sift_amd.texture import TextureContext
    ctx = TextureContext(0)
    try:
        p = ctx.default_params()
        p.gray_shift = _cv2_gray_shift(cv2)
        for kind, f in _frames():
            gray, codes, edges = ctx.debug_planes(f, p)
            want_gray = cv2.cvtColor(f, cv2.COLOR_BGR2GRAY)
            assert np.array_equal(gray, want_gray), ("gray", kind, f.shape)
            assert np.array_equal(edges, cv2.Canny(want_gray, 50, 150)), ("Canny", kind, f.shape)
            lbp = feature.local_binary_pattern(want_gray, 8, 1, method="uniform").astype(np.uint8)
            assert np.array_equal(codes, lbp), ("lbp", kind, f.shape)
            r = ctx.analyze_frames([f], p)[0]
            hp = cv2.filter2D(want_gray, -1, HP_KERNEL).astype(np.int64)
            assert (r.hp_sum, r.hp_sumsq) == (int(hp.sum()), int((hp * hp).sum())), ("filter2D", kind, f.shape)
            ad = cv2.absdiff(want_gray, cv2.GaussianBlur(want_gray, (5, 5), 0)).astype(np.int64)
            assert r.absdiff_sum == int(ad.sum()), ("GaussianBlur", kind, f.shape)
            freq = float(np.std(np.log(np.abs(np.fft.fftshift(np.fft.fft2(want_gray))) + 1)))
            assert r.frequency_anomaly == pytest.approx(freq, rel=1e-4, abs=1e-9), ("fft", kind, f.shape)
    finally:
        ctx.close()
