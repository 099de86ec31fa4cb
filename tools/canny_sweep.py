"""Dense against sparse placement of Canny's suppression as the flooded set grows: a 150-frame 1080p clip analysed with
lower and lower thresholds (more candidates, more strong pixels, a larger flooded set), milliseconds per clip in each
mode and the flooded fraction of the first frame (computed on the host with scipy).  Run on the GPU box from the
repository root: python3 tools/canny_sweep.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from scipy import ndimage as ndi

from oracle import texture_oracle as O
from sift_amd._lib import StxMetrics
from sift_amd.synth import KIND_ID, synth_frame
from sift_amd.texture import TextureContext

nf, h, w = 150, 1080, 1920
seed = 0x5EED0002
ctxs = {}
for mode in ("dense", "sparse"):
    ctxs[mode] = TextureContext(0)
    ctxs[mode].set_canny_mode(mode)
t = torch.empty(nf * h * w * 3, dtype=torch.uint8, device="cuda")
ctxs["dense"].synth_device(t.data_ptr(), KIND_ID["natural"], seed, 0, 0, nf, h, w)
torch.cuda.synchronize()
g = O.COracle().gray(synth_frame("natural", seed, 0, 0, h, w)).astype(np.int32)
gp = np.pad(g, 1, mode="edge")
gx = (gp[:-2, 2:] - gp[:-2, :-2]) + 2 * (gp[1:-1, 2:] - gp[1:-1, :-2]) + (gp[2:, 2:] - gp[2:, :-2])
gy = (gp[2:, :-2] - gp[:-2, :-2]) + 2 * (gp[2:, 1:-1] - gp[:-2, 1:-1]) + (gp[2:, 2:] - gp[:-2, 2:])
mag = np.abs(gx) + np.abs(gy)
out = (StxMetrics * nf)()
for low, high in ((50, 150), (45, 130), (40, 115), (35, 100), (30, 90), (25, 80)):
    cand, strong = mag > low, mag > high
    lab, _ = ndi.label(cand, structure=np.ones((3, 3)))
    keep = np.unique(lab[strong])
    R = np.isin(lab, keep[keep > 0]).mean()
    res = {}
    for mode, c in ctxs.items():
        p = c.default_params()
        p.canny_low, p.canny_high = low, high
        for _ in range(3):
            c.analyze_device(t.data_ptr(), nf, h, w, params=p, out=out)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 8
        pend = None
        outs = [(StxMetrics * nf)(), (StxMetrics * nf)()]
        for i in range(n):
            tk = c.submit_device(t.data_ptr(), nf, h, w, outs[i % 2], p)
            if pend is not None:
                c.wait(pend)
            pend = tk
        c.wait(pend)
        res[mode] = (time.perf_counter() - t0) / n * 1e3
        res[mode + "_edges"] = outs[(n - 1) % 2][0].edge_count
    print(f"low {low} high {high}: candidates {cand.mean():.3f} strong {strong.mean():.4f} flooded {R:.4f}  "
          f"dense {res['dense']:.3f} ms  sparse {res['sparse']:.3f} ms  edges equal {res['dense_edges'] == res['sparse_edges']}", flush=True)
