"""The callers of the texture path (sift_amd/checkers.py): scalar fusion on the CPU, per-frame quality
statistics and whole-clip verdicts on the GPU against the oracle's independent restatement.

Pins: OpenCV is absent here, so the Laplacian / variance inputs are "parity unpinned" as far as the
reference is concerned; the C and numpy/scipy restatements must agree bit for bit and the device
sums must equal them.
"""
import numpy as np
import pytest

from oracle.texture_oracle import (COracle, np_fallback_check, np_gray, np_improved_fallback_base, np_quality_sums,
                                   np_quality_variances)
from sift_amd.checkers import fuse_passive_scores
from sift_amd.synth import synth_frame

SEED = 0x5EED0000


def test_quality_sums_c_and_numpy_agree():
    orc = COracle()
    rng = np.random.default_rng(7)
    for shape in [(1, 1), (1, 9), (7, 1), (2, 2), (5, 3), (33, 70), (128, 131)]:
        g = rng.integers(0, 256, shape, dtype=np.uint8)
        assert orc.quality_sums(g) == np_quality_sums(g)
    flat = np.full((12, 12), 77, np.uint8)
    assert orc.quality_sums(flat) == (0, 0, 144 * 77 * 77)
    # hand-computed: a single bright pixel in the middle of a 3x3 zero image
    dot = np.zeros((3, 3), np.uint8)
    dot[1, 1] = 10
    s, q, g2 = orc.quality_sums(dot)
    # Laplacian image: centre -40; each of the four edge-neighbours sees the dot once directly and once
    # more through the reflect-101 border (+20); corners see nothing
    assert (s, q, g2) == (-40 + 4 * 20, 1600 + 4 * 400, 100)


def _res(avg, passed=True, key="average_score"):
    return {key: avg, "passed": passed}


def test_fuse_passive_scores_rules():
    tex = _res(0.7)
    ident = _res(0.9, key="consistency_score")
    cam = {"risk_score": 0.1, "passed": True}
    r = fuse_passive_scores([0.8, 0.6], tex, ident, cam)
    w = 0.7 * 0.4 + 0.7 * 0.25 + 0.9 * 0.25 + 0.9 * 0.1
    assert r["passed"] is True and r["message"] == "Enhanced passive check passed"
    assert r["average_real_score"] == round(w, 3) and r["base_model_score"] == 0.7
    assert r["frame_scores"] == [0.8, 0.6]
    assert list(r) == ["passed", "average_real_score", "base_model_score", "frame_scores", "message", "enhanced_analysis"]
    assert list(r["enhanced_analysis"]) == ["texture_analysis", "identity_analysis", "virtual_camera_analysis",
                                            "weighted_scores", "weights_used", "critical_failures"]
    assert r["enhanced_analysis"]["critical_failures"] == []
    # one failed detector vetoes a combined score below 0.8
    r = fuse_passive_scores([0.8, 0.6], _res(0.7, passed=False), ident, cam)
    assert r["passed"] is False and r["message"] == "Enhanced passive check failed: texture_analysis"
    # ... but not a very high one
    r = fuse_passive_scores([1.0], _res(0.9, passed=False), _res(1.0, key="consistency_score"),
                            {"risk_score": 0.0, "passed": True})
    assert r["passed"] is True and r["enhanced_analysis"]["critical_failures"] == ["texture_analysis"]
    # no failures, low score
    r = fuse_passive_scores([], _res(0.5), _res(0.5, key="consistency_score"), {"risk_score": 0.5, "passed": True})
    assert r["passed"] is False and r["base_model_score"] == 0.0
    assert r["message"] == "Enhanced passive check failed (weighted score: 0.300)"


# ------------------------------------------------------------------------------------------- GPU
def _frames(kind, n, h, w, video=3):
    return [synth_frame(kind, SEED + 9, video, f, h, w) for f in range(n)]


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(1, 1), (1, 40), (37, 1), (5, 7), (64, 64), (65, 130), (200, 333)])
@pytest.mark.parametrize("kind", ["noise", "natural", "flat"])
def test_quality_sums_device_exact(shape, kind):
    from sift_amd.texture import TextureContext
    ctx = TextureContext(0)
    orc = COracle()
    h, w = shape
    frames = _frames(kind, 3, h, w)
    p = ctx.default_params()
    p.quality_stats = 1
    res = ctx.analyze_frames(frames, p)
    for f, m in zip(frames, res):
        g = np_gray(f)
        s, q, g2 = orc.quality_sums(g)
        assert (m.lap_sum, m.lap_sumsq, m.gray_sumsq) == (s, q, g2)
        lv, bv = np_quality_variances(g)
        assert m.laplacian_variance == pytest.approx(lv, rel=1e-9, abs=1e-9)
        assert m.brightness_variance == pytest.approx(bv, rel=1e-9, abs=1e-9)
    # the quality pass changes nothing else, and is off by default
    plain = ctx.analyze_frames(frames)
    for a, b in zip(res, plain):
        assert a.texture_score == b.texture_score and a.edge_count == b.edge_count
        assert (b.lap_sum, b.lap_sumsq, b.gray_sumsq, b.laplacian_variance) == (0, 0, 0, 0.0)


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["noise", "natural", "step"])
def test_fallback_checker_matches_restatement(kind):
    from sift_amd.checkers import FallbackPassiveChecker
    chk = FallbackPassiveChecker()
    assert chk.threshold == 0.5
    frames = _frames(kind, 12, 72, 96)
    got = chk.check(frames)
    want = np_fallback_check(frames)
    assert list(got) == ["passed", "average_real_score", "message"]
    assert got == want
    assert chk.check([]) == {"passed": False, "average_real_score": 0.0, "message": "No frames to analyze"}
    # a frame the path cannot read scores the reference's default 0.5
    bad = chk.check([np.zeros((4, 4), np.uint8)])
    assert bad["average_real_score"] == 0.5 and bad["passed"] is True


class _Identity:
    def analyze_face_consistency(self, frames):
        return {"passed": True, "consistency_score": 0.85, "message": "stub"}


class _Camera:
    def detect_virtual_camera(self):
        return {"passed": True, "risk_score": 0.2, "message": "stub"}


class _Broken:
    def analyze_face_consistency(self, frames):
        raise RuntimeError("no faces")


@pytest.mark.gpu
def test_improved_fallback_checker():
    from sift_amd.checkers import ImprovedFallbackPassiveChecker
    from sift_amd.texture import TextureNoiseDetector
    frames = _frames("natural", 20, 72, 96)
    basic = ImprovedFallbackPassiveChecker()
    assert basic.threshold == 0.3 and basic.has_face_detector is False and basic.enhanced_available is False
    r = basic.check(frames)
    base = np_improved_fallback_base(frames)
    assert list(r) == ["passed", "average_real_score", "base_score", "face_detection_rate", "message", "enhanced_analysis"]
    assert r["base_score"] == round(base, 3) and r["average_real_score"] == round(base, 3)
    assert r["enhanced_analysis"] is None and r["face_detection_rate"] == 0
    assert r["message"] == f"Basic fallback check ({'passed' if base >= 0.3 else 'failed'})"
    assert basic.check([])["message"] == "No frames to analyze"

    full = ImprovedFallbackPassiveChecker(identity_detector=_Identity(), virtual_camera_detector=_Camera())
    r = full.check(frames)
    tex = TextureNoiseDetector().compute_texture_score(frames)
    assert r["enhanced_analysis"]["texture_analysis"] == tex  # reused records == a separate pass
    want = base * 0.5 + tex["average_score"] * 0.2 + 0.85 * 0.2 + 0.8 * 0.1
    assert r["average_real_score"] == round(want, 3)
    assert r["enhanced_analysis"]["weights_used"] == {"base": 0.5, "texture": 0.2, "identity": 0.2, "virtual_camera": 0.1}
    assert r["message"].startswith("Enhanced fallback check")

    # a failing detector degrades to the basic score (main.py:292-294)
    degraded = ImprovedFallbackPassiveChecker(identity_detector=_Broken(), virtual_camera_detector=_Camera()).check(frames)
    assert degraded["enhanced_analysis"] is None and degraded["average_real_score"] == round(base, 3)
