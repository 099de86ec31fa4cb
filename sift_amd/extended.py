"""Multi-scale additions to the texture path, kept apart from the drop-in class.

Nothing here exists in the reference: uniform-LBP histograms at (P, R) = (8, 1), (16, 2), (24, 3),
Sobel gradient statistics on the images ``cv2.Canny`` builds at
/root/reference/app/passive_checker.py:116 and an 8x8 blockiness measure.  ``analyze_frames`` returns,
per frame, the reference's five metrics (passive_checker.py:125-131) unchanged under ``"metrics"`` and
the additions under their own keys, computed in the same device pass.
"""
from __future__ import annotations

from typing import Any, Dict, List, Optional, Sequence

import numpy as np

from ._lib import StxExtended, StxMetrics, StxParams
from .texture import TextureContext, TextureNoiseDetector, metrics_dict

EXTENDED_FLOAT_KEYS = ("sobel_mean", "sobel_std", "strong_gradient_fraction", "edge_consistency", "blockiness")


def extended_dict(m: StxMetrics, x: StxExtended) -> Dict[str, Any]:
    return {
        "metrics": metrics_dict(m),
        "texture_score": float(m.texture_score),
        "lbp_histograms": {
            "P8_R1": [int(v) for v in m.lbp_hist[:10]],
            "P16_R2": [int(v) for v in x.lbp_hist_p16r2[:18]],
            "P24_R3": [int(v) for v in x.lbp_hist_p24r3[:26]],
        },
        **{k: float(getattr(x, k)) for k in EXTENDED_FLOAT_KEYS},
        "sobel_sum": int(x.sobel_sum),
        "sobel_sumsq": int(x.sobel_sumsq),
        "sobel_strong": int(x.sobel_strong),
        "block_sums": [int(v) for v in x.block_sums],
        "block_counts": [int(v) for v in x.block_counts],
    }


class ExtendedTextureAnalyzer:
    def __init__(self, device: int = 0, params: Optional[StxParams] = None):
        self._ctx = TextureContext(device)
        self._params = params

    def analyze_frames(self, frames: Sequence[np.ndarray]) -> List[Dict[str, Any]]:
        frames = [TextureNoiseDetector._as_frame(f) for f in frames]
        if not frames:
            return []
        out, ext = self._ctx.analyze_frames_extended(frames, self._params)
        return [extended_dict(out[i], ext[i]) for i in range(len(frames))]
