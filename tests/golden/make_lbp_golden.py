"""Generate the LBP goldens with scikit-image (the library the reference calls at
/root/reference/app/passive_checker.py:103).

Run with the interpreter that has scikit-image:
    /opt/conda/bin/python3.9 tests/golden/make_lbp_golden.py
Writes tests/golden/lbp_skimage.npz: one uint8 code image per (image, P, R),
plus the rounded sample-offset tables numpy produced.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
sys.path.insert(0, HERE)

import skimage  # noqa: E402
from skimage.feature import local_binary_pattern  # noqa: E402

import hashlib  # noqa: E402

from cases import LBP_IMAGES, LBP_PR, PATCHWORK_CASES, SEED  # noqa: E402
from patchwork import patchwork_gray  # noqa: E402
from sift_amd.synth import synth_frame  # noqa: E402
from lattice import lattice_image  # noqa: E402

out = {"skimage_version": np.array(skimage.__version__), "numpy_version": np.array(np.__version__)}
for (P, R) in LBP_PR:
    ang = 2 * np.pi * np.arange(P, dtype=np.double) / P
    out["rp_P%d" % P] = np.round(-R * np.sin(ang), 5)
    out["cp_P%d" % P] = np.round(R * np.cos(ang), 5)
for idx, (kind, h, w) in enumerate(LBP_IMAGES):
    gray = np.ascontiguousarray(synth_frame(kind, SEED, 0, idx, h, w)[..., 1])
    for (P, R) in LBP_PR:
        codes = local_binary_pattern(gray, P=P, R=R, method="uniform")
        assert codes.dtype == np.float64 and codes.min() >= 0 and codes.max() <= P + 1
        assert np.all(codes == np.floor(codes))
        out["lbp_%02d_%s_%dx%d_P%d" % (idx, kind, h, w, P)] = codes.astype(np.uint8)
lat = lattice_image()
for (P, R) in LBP_PR:
    out["lbp_lattice_P%d" % P] = local_binary_pattern(lat, P=P, R=R, method="uniform").astype(np.uint8)
for (h, w, cell) in PATCHWORK_CASES:
    codes = local_binary_pattern(patchwork_gray(h, w, cell), P=8, R=1.0, method="uniform").astype(np.uint8)
    out["patch_%dx%d_c%d_hist" % (h, w, cell)] = np.bincount(codes.ravel(), minlength=10).astype(np.int64)
    out["patch_%dx%d_c%d_sha256" % (h, w, cell)] = np.array(hashlib.sha256(codes.tobytes()).hexdigest())
np.savez_compressed(os.path.join(HERE, "lbp_skimage.npz"), **out)
print("wrote", len(out), "arrays; skimage", skimage.__version__)
