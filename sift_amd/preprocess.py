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
        out = torch.empty((len(frames), 3, h, w), dtype=torch.float32, device=self.device)
        if not frames:
            return out
        if len({f.shape for f in frames}) != 1:
            raise ValueError("all frames of one call must have the same shape")
        # frames go up once, the kernel writes the network's input tensor in place: nothing comes back to the host
        src = torch.from_numpy(np.ascontiguousarray(np.stack(frames))).to(self.device)
        fh, fw = frames[0].shape[:2]
        self._ctx.preprocess_device(src.data_ptr(), len(frames), fh, fw, out.data_ptr(), out_h=h, out_w=w)
        return out

    def _preprocess_frame(self, frame: np.ndarray) -> torch.Tensor:
        return self.preprocess_frames([frame])
