"""The immediate callers of the texture path, fed from the same device pass.

The reference scores a clip in ``PassiveChecker.check``
(/root/reference/app/passive_checker.py:445-559) or, without model weights, in
``FallbackPassiveChecker.check`` / ``ImprovedFallbackPassiveChecker.check``
(/root/reference/app/main.py:97-141, :143-305).  All three are scalar glue around per-frame image
statistics; the two fallback checkers recompute gray, a Laplacian variance, a gray variance and
``Canny(50,150)`` per frame on the host.  Here those statistics come out of the pass that
``TextureNoiseDetector`` runs anyway (``stx_params.quality_stats``), so a clip is read once.

Kept from the reference: class names, ``threshold`` values, how many frames are looked at, every
weight and clamp, rounding, dict keys and message strings.  Not part of this path and therefore
injected by the caller when wanted: the CNN frame scores (``fuse_passive_scores`` takes them as
numbers), the face-identity and virtual-camera detectors (any objects with the reference's method
names), and the Haar face cascade (``has_face_detector`` is False, which is the reference's own
behaviour when the cascade cannot be loaded, main.py:157-160).
"""
from __future__ import annotations

import ctypes as C
import logging
from typing import Any, Dict, List, Optional, Sequence

import numpy as np

from ._lib import StxMetrics
from .texture import TextureNoiseDetector

logger = logging.getLogger(__name__)


def _quality_records(detector: TextureNoiseDetector, frames: Sequence[np.ndarray]) -> List[Optional[StxMetrics]]:
    """One record per frame with the quality fields filled; None where the frame cannot be analysed
    (the reference catches per-frame errors and substitutes a default score)."""
    ctx = detector._ctx
    # a private copy of the detector's parameters: the detector (and anyone holding its struct) never sees the change
    params = ctx.default_params()
    if detector._params is not None:
        C.memmove(C.byref(params), C.byref(detector._params), C.sizeof(params))
    params.quality_stats = 1
    out: List[Optional[StxMetrics]] = [None] * len(frames)
    groups: Dict[tuple, List[int]] = {}
    arrs: List[Optional[np.ndarray]] = []
    for i, f in enumerate(frames):
        try:
            a = detector._as_frame(f)
        except (TypeError, ValueError) as e:
            logger.warning("Error processing frame: %s", e)
            arrs.append(None)
            continue
        arrs.append(a)
        groups.setdefault(a.shape, []).append(i)
    for idxs in groups.values():
        res = ctx.analyze_frames([arrs[i] for i in idxs], params)
        for j, i in enumerate(idxs):
            r = StxMetrics()
            C.memmove(C.byref(r), C.byref(res[j]), C.sizeof(StxMetrics))
            out[i] = r
    return out


class FallbackPassiveChecker:
    """main.py:97-141: blur and brightness of the first ten frames."""

    def __init__(self, device: int = 0, texture_detector: Optional[TextureNoiseDetector] = None):
        self.threshold = 0.5  # main.py:103
        self._detector = texture_detector or TextureNoiseDetector(device)

    def check(self, frames) -> Dict[str, Any]:
        if not frames:
            return {"passed": False, "average_real_score": 0.0, "message": "No frames to analyze"}
        scores = []
        for rec in _quality_records(self._detector, list(frames[:10])):
            if rec is None:
                scores.append(0.5)  # main.py:133
                continue
            sharp = min(rec.laplacian_variance / 1000.0, 1.0)
            varied = min(rec.brightness_variance / 10000.0, 1.0)
            scores.append((sharp + varied) / 2.0)
        average = sum(scores) / len(scores) if scores else 0.5
        passed = bool(average >= self.threshold)
        return {
            "passed": passed,
            "average_real_score": round(float(average), 3),
            "message": "Fallback passive check (basic image quality analysis)",
        }


class ImprovedFallbackPassiveChecker:
    """main.py:143-305: image quality of the first fifteen frames, fused with the enhanced detectors.

    ``identity_detector`` / ``virtual_camera_detector``: optional objects with the reference's
    ``analyze_face_consistency(frames)`` / ``detect_virtual_camera()`` methods.  The enhanced fusion runs
    only when both are given (the reference enables it only when all three detectors import, main.py:162-174).
    """

    def __init__(self, device: int = 0, identity_detector=None, virtual_camera_detector=None,
                 texture_detector: Optional[TextureNoiseDetector] = None):
        self.threshold = 0.3  # main.py:149
        self.has_face_detector = False  # the Haar cascade is not part of this path (main.py:157-160)
        self.texture_detector = texture_detector or TextureNoiseDetector(device)
        self.identity_detector = identity_detector
        self.virtual_camera_detector = virtual_camera_detector
        self.enhanced_available = identity_detector is not None and virtual_camera_detector is not None

    def check(self, frames) -> Dict[str, Any]:
        if not frames:
            return {"passed": False, "average_real_score": 0.0, "message": "No frames to analyze",
                    "enhanced_analysis": None}
        head = list(frames[:15])
        records = _quality_records(self.texture_detector, head)
        scores = []
        for rec in records:
            if rec is None:
                scores.append(0.3)  # main.py:230
                continue
            # no face was looked for, so the weighting is the reference's no-face branch (main.py:223-224)
            sharp = min(rec.laplacian_variance / 500.0, 1.0)
            varied = min(rec.brightness_variance / 5000.0, 1.0)
            edgy = min(rec.edge_density * 20, 1.0)
            scores.append(sharp * 0.4 + varied * 0.3 + edgy * 0.3)
        if not scores:
            return {"passed": False, "average_real_score": 0.0, "message": "Failed to process any frames",
                    "enhanced_analysis": None}
        base = sum(scores) / len(scores)
        face_detection_rate = 0  # main.py:241 without a face detector
        base = min(base, 1.0)

        enhanced_analysis = None
        final = base
        if self.enhanced_available:
            try:
                # the first ten frames were analysed above: reuse their records when all of them are usable
                first = records[:10]
                if all(r is not None for r in first):
                    texture_result = self.texture_detector.result_from_records(first)  # type: ignore[arg-type]
                else:
                    texture_result = self.texture_detector.compute_texture_score(frames)
                identity_result = self.identity_detector.analyze_face_consistency(frames)
                virtual_camera_result = self.virtual_camera_detector.detect_virtual_camera()
                weights = {"base": 0.5, "texture": 0.2, "identity": 0.2, "virtual_camera": 0.1}
                parts = {
                    "base": base,
                    "texture": texture_result["average_score"],
                    "identity": identity_result["consistency_score"],
                    "virtual_camera": 1.0 - virtual_camera_result["risk_score"],
                }
                final = sum(parts[k] * weights[k] for k in weights)
                enhanced_analysis = {
                    "texture_analysis": texture_result,
                    "identity_analysis": identity_result,
                    "virtual_camera_analysis": virtual_camera_result,
                    "enhanced_scores": {k: round(v, 3) for k, v in parts.items()},
                    "weights_used": weights,
                }
            except Exception as e:  # main.py:292-294: degrade to the basic score
                logger.warning("Enhanced analysis failed, falling back to basic analysis: %s", e)
                final = base
                enhanced_analysis = None
        passed = bool(final >= self.threshold)
        return {
            "passed": passed,
            "average_real_score": round(float(final), 3),
            "base_score": round(float(base), 3),
            "face_detection_rate": round(face_detection_rate, 3),
            "message": f"{'Enhanced' if enhanced_analysis else 'Basic'} fallback check "
                       f"({'passed' if passed else 'failed'})",
            "enhanced_analysis": enhanced_analysis,
        }


def fuse_passive_scores(frame_scores: Sequence[float], texture_result: Dict[str, Any],
                        identity_result: Dict[str, Any], virtual_camera_result: Dict[str, Any]) -> Dict[str, Any]:
    """The verdict of ``PassiveChecker.check`` (passive_checker.py:479-559) from its ingredients:
    ``frame_scores`` are the per-frame "real" probabilities of the anti-spoofing CNN (possibly empty), the
    three dicts the results of the three detectors.  (The reference also compares the CNN mean with the
    checker's threshold, passive_checker.py:484, but never uses the outcome.)"""
    scores = [float(s) for s in frame_scores]
    base_score = float(np.mean(scores)) if scores else 0.0
    weights = {"base_model": 0.4, "texture": 0.25, "identity": 0.25, "virtual_camera": 0.1}
    normalized = {
        "base_model": base_score,
        "texture": texture_result["average_score"],
        "identity": identity_result["consistency_score"],
        "virtual_camera": 1.0 - virtual_camera_result["risk_score"],
    }
    weighted = sum(normalized[k] * weights[k] for k in weights)
    failures = []
    if not texture_result["passed"]:
        failures.append("texture_analysis")
    if not identity_result["passed"]:
        failures.append("identity_consistency")
    if not virtual_camera_result["passed"]:
        failures.append("virtual_camera_detected")
    # one failed detector vetoes anything short of a very high combined score (passive_checker.py:532-541)
    if failures and weighted < 0.8:
        passed, message = False, f"Enhanced passive check failed: {', '.join(failures)}"
    elif weighted >= 0.6:
        passed, message = True, "Enhanced passive check passed"
    else:
        passed, message = False, f"Enhanced passive check failed (weighted score: {weighted:.3f})"
    return {
        "passed": bool(passed),
        "average_real_score": round(float(weighted), 3),
        "base_model_score": round(float(base_score), 3),
        "frame_scores": [round(float(s), 3) for s in scores],
        "message": str(message),
        "enhanced_analysis": {
            "texture_analysis": texture_result,
            "identity_analysis": identity_result,
            "virtual_camera_analysis": virtual_camera_result,
            "weighted_scores": {k: round(float(v), 3) for k, v in normalized.items()},
            "weights_used": weights,
            "critical_failures": failures,
        },
    }
