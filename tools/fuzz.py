import sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, 'tests/golden')
import numpy as np
from oracle import texture_oracle as O
from sift_amd.texture import TextureContext
from sift_amd.synth import synth_frame, KINDS
ctx = TextureContext(0)
c = O.COracle()
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
bad = 0
N = int(sys.argv[2]) if len(sys.argv) > 2 else 150
for it in range(N):
    big = rng.random() < 0.15
    h = int(rng.integers(1, 1300 if big else 400)); w = int(rng.integers(1, 2200 if big else 500))
    if rng.random() < 0.6:  # widths the stencil pass converts itself
        w = max(4, w & ~3)
    kind = KINDS[int(rng.integers(0, len(KINDS)))]
    n = int(rng.integers(1, 4))
    frames = [synth_frame(kind, 0x5EED0000 + it, it, i, h, w) for i in range(n)]
    if rng.random() < 0.3:  # posterize: flat patches
        q = int(rng.integers(2, 6))
        frames = [np.ascontiguousarray((f >> q) << q) for f in frames]
    p = ctx.default_params()
    p.quality_stats = 1 if rng.random() < 0.5 else 0
    res = ctx.analyze_frames(frames, p)
    if it % 25 == 0:
        print("case", it, flush=True)
    for f, r in zip(frames, res):
        m = c.analyze_frame(f)
        if p.quality_stats and (r.lap_sum, r.lap_sumsq, r.gray_sumsq) != c.quality_sums(c.gray(f)):
            bad += 1
            print("QUALITY MISMATCH", it, kind, h, w, flush=True)
        a = (tuple(r.lbp_hist[:10]), r.edge_count, r.absdiff_sum, r.hp_sum, r.hp_sumsq, r.gray_sum)
        b = (tuple(m.lbp_hist[:10]), m.edge_count, m.absdiff_sum, m.hp_sum, m.hp_sumsq, m.gray_sum)
        fa = abs(r.frequency_anomaly - m.frequency_anomaly) <= 1e-4 * max(abs(m.frequency_anomaly), 1e-9) + 1e-9
        if a != b or not fa:
            bad += 1
            print("MISMATCH", it, kind, h, w, a, b, r.frequency_anomaly, m.frequency_anomaly, flush=True)
print("fuzz done", N, "cases, mismatches:", bad)
