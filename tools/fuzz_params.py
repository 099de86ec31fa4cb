"""Random frames with random parameters (LBP points / radius, Canny thresholds, gray coefficient set, FFT crop and
precision, Canny placement, chunking, quality statistics) through the HIP path against the C oracle.
Run on the GPU box from the repository root: python3 tools/fuzz_params.py [seed] [cases]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from oracle import texture_oracle as O
from sift_amd.synth import KINDS, synth_frame
from sift_amd.texture import TextureContext

ctx = TextureContext(0)
c = O.COracle()
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 100
bad = 0
LBP = [(8, 1.0), (8, 1.0), (16, 2.0), (24, 3.0), (5, 1.5), (12, 2.5), (8, 2.0), (4, 1.0)]
for it in range(N):
    h = int(rng.integers(8, 500)); w = int(rng.integers(8, 700))
    if rng.random() < 0.6:
        w = max(8, w & ~3)
    kind = KINDS[int(rng.integers(0, len(KINDS)))]
    n = int(rng.integers(1, 5))
    frames = [synth_frame(kind, 0x5EED0100 + it, it, i, h, w) for i in range(n)]
    if rng.random() < 0.25:
        q = int(rng.integers(2, 6))
        frames = [np.ascontiguousarray((f >> q) << q) for f in frames]
    p, op = ctx.default_params(), c.default_params()
    P, R = LBP[int(rng.integers(0, len(LBP)))]
    lo, hi = int(rng.integers(0, 300)), int(rng.integers(0, 600))
    shift = 14 if rng.random() < 0.4 else 15
    fh = int(rng.integers(1, h + 1)) if rng.random() < 0.4 else 0
    fw = int(rng.integers(1, w + 1)) if fh else 0
    exact = int(rng.integers(-1, 2))
    for q_ in (p, op):
        q_.lbp_points, q_.lbp_radius, q_.canny_low, q_.canny_high, q_.gray_shift = P, R, lo, hi, shift
        q_.fft_h, q_.fft_w = fh, fw
    p.fft_exact = exact
    p.quality_stats = 1 if rng.random() < 0.3 else 0
    ctx.set_canny_mode(("dense", "sparse", "auto")[int(rng.integers(0, 3))])
    ctx.set_chunk_frames(int(rng.integers(0, 3)))
    res = ctx.analyze_frames(frames, p)
    if it % 20 == 0:
        print("case", it, flush=True)
    for f, r in zip(frames, res):
        m = c.analyze_frame(f, op)
        a = (tuple(r.lbp_hist[:P + 2]), r.edge_count, r.absdiff_sum, r.hp_sum, r.hp_sumsq, r.gray_sum)
        b = (tuple(m.lbp_hist[:P + 2]), m.edge_count, m.absdiff_sum, m.hp_sum, m.hp_sumsq, m.gray_sum)
        # fp32-only transforms of content with exactly-zero bins are allowed their noise floor
        tol = 1e-4 if exact >= 0 else 5e-2
        fa = abs(r.frequency_anomaly - m.frequency_anomaly) <= tol * max(abs(m.frequency_anomaly), 1e-9) + 1e-9
        if a != b or not fa:
            bad += 1
            print("MISMATCH", it, kind, h, w, (P, R, lo, hi, shift, fh, fw, exact), a == b, r.frequency_anomaly, m.frequency_anomaly, flush=True)
print("fuzz_params done", N, "cases, mismatches:", bad)
