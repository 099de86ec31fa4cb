"""The adversarial near-tie image of the LBP goldens (see cases.LATTICE_ALPHABET)."""
import numpy as np

from cases import LATTICE_ALPHABET, LATTICE_SHAPE, SEED
from sift_amd.synth import pixel_hash


def lattice_image() -> np.ndarray:
    h, w = LATTICE_SHAPE
    hs = pixel_hash(SEED + 77, 5, 0, h, w)
    idx = (hs >> np.uint64(17)) % np.uint64(len(LATTICE_ALPHABET))
    return np.asarray(LATTICE_ALPHABET, dtype=np.uint8)[idx.astype(np.int64)]
