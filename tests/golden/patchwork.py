"""Patchwork images for the flat-region LBP goldens (pure numpy integer arithmetic: identical under every
numpy version)."""
import numpy as np

from sift_amd.synth import pixel_hash


def patchwork_gray(h: int, w: int, cell: int, seed: int = 0x5EED0700) -> np.ndarray:
    v = (pixel_hash(seed, 3, 0, (h + cell - 1) // cell, (w + cell - 1) // cell) & np.uint64(0xFF)).astype(np.uint8)
    return np.ascontiguousarray(np.repeat(np.repeat(v, cell, axis=0), cell, axis=1)[:h, :w])
