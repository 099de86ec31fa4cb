"""Frame preparation of the reference's CNN branch on the GPU.

``FramePreprocessor._preprocess_frame`` mirrors PassiveChecker._preprocess_frame
(/root/reference/app/passive_checker.py:393-418): BGR -> RGB, cv2.resize to ``input_size``
(80 x 80, :355) with the default bilinear interpolation, float32 / 255, channels first, a batch
dimension in front, as a torch tensor on ``device``.  ``preprocess_frames`` does the frames of
the loop at :470-474 in one launch.
"""
from __future__ import annotations

from typing import Sequence, Tuple

import numpy as np
import torch

from .texture import TextureContext, TextureNoiseDetector


class FramePreprocessor:
    def __init__(self, device: int = 0, input_size: Tuple[int, int] = (80, 80)):
        self.input_size = tuple(input_size)  # (width, height), as cv2.resize takes it
        self.device = torch.device("cuda", device)
        self._ctx = TextureContext(device)

    def preprocess_frames(self, frames: Sequence[np.ndarray]) -> torch.Tensor:
        """[N][3][H][W] float32 on the GPU from N same-shaped BGR frames."""
        frames = [TextureNoiseDetector._as_frame(f) for f in frames]
        w, h = self.input_size
        out = self._ctx.preprocess_frames(frames, out_h=h, out_w=w)
        return torch.from_numpy(out).to(self.device)

    def _preprocess_frame(self, frame: np.ndarray) -> torch.Tensor:
        return self.preprocess_frames([frame])
