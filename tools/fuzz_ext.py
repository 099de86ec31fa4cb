"""Random frames through the extended statistics (Sobel / blockiness sums, LBP at (16,2) and (24,3)) against the
numpy restatement.  Run on the GPU box from the repository root: python3 tools/fuzz_ext.py [seed] [cases]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from oracle.texture_oracle import COracle, np_extended_sums, np_gray
from sift_amd.extended import ExtendedTextureAnalyzer
from sift_amd.synth import KINDS, synth_frame

ana = ExtendedTextureAnalyzer()
c = COracle()
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 100
bad = 0
for it in range(N):
    big = rng.random() < 0.1
    h = int(rng.integers(1, 700 if big else 200))
    w = int(rng.integers(1, 1100 if big else 300))
    if rng.random() < 0.5:
        w = max(4, w & ~3)
    kind = KINDS[int(rng.integers(0, len(KINDS)))]
    frames = [synth_frame(kind, 0x5EED0000 + it, it, i, h, w) for i in range(int(rng.integers(1, 3)))]
    if rng.random() < 0.3:
        q = int(rng.integers(2, 6))
        frames = [np.ascontiguousarray((f >> q) << q) for f in frames]
    res = ana.analyze_frames(frames)
    if it % 20 == 0:
        print("case", it, flush=True)
    for f, r in zip(frames, res):
        g = np_gray(f)
        want = np_extended_sums(g)
        ok = all(r[k] == want[k] for k in ("sobel_sum", "sobel_sumsq", "sobel_strong", "block_sums", "block_counts"))
        for key, P, R in (("P16_R2", 16, 2.0), ("P24_R3", 24, 3.0)):
            codes, _ = c.lbp(g, P, R)
            ok = ok and r["lbp_histograms"][key] == np.bincount(codes.ravel().astype(np.int64), minlength=P + 2).tolist()
        if not ok:
            bad += 1
            print("MISMATCH", it, kind, h, w, flush=True)
print("fuzz_ext done", N, "cases, mismatches:", bad)
