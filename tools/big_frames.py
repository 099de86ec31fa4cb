"""Frames at the size limits (8192 x 8192, one-word-wide, three-row, 8K UHD) against the CPU oracle.  Run on the GPU box from the repository root."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import texture_oracle as O
from sift_amd.texture import TextureContext
from sift_amd.synth import synth_frame
ctx = TextureContext(0)
c = O.COracle()
for (h, w) in [(8192, 8192), (8192, 4), (3, 8192), (4320, 7680)]:
    f = synth_frame("natural", 0x5EED0000, 3, 0, h, w)
    t = time.time(); r = ctx.analyze_frames([f])[0]; tg = time.time() - t
    t = time.time(); m = c.analyze_frame(f); tc = time.time() - t
    a = (tuple(r.lbp_hist[:10]), r.edge_count, r.absdiff_sum, r.hp_sum, r.hp_sumsq, r.gray_sum)
    b = (tuple(m.lbp_hist[:10]), m.edge_count, m.absdiff_sum, m.hp_sum, m.hp_sumsq, m.gray_sum)
    print(h, w, "ints equal:", a == b, "freq rel:", abs(r.frequency_anomaly - m.frequency_anomaly) / max(abs(m.frequency_anomaly), 1e-9), "gpu s", round(tg, 3), "cpu s", round(tc, 1), flush=True)
