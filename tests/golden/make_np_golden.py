"""Generate the numpy-side goldens (numpy is the library the reference calls for
fft2 / std / var / mean at /root/reference/app/passive_checker.py:104-123).

Run with the main interpreter (numpy 2.2.6):
    python tests/golden/make_np_golden.py
Writes
  tests/golden/fft_numpy.json    std / sum / sumsq of log(|fft2(gray)|+1) per image
  tests/golden/score_sets.json, score_sets_gray14.json
                                 per-frame metrics and compute_texture_score dicts (15-bit / 14-bit gray weights)
                                 for small frame sets.  LBP and FFT parts come from
                                 algorithms pinned by scikit-image / numpy; the OpenCV
                                 stages come from the double-implemented restatement
                                 (parity unpinned).
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
sys.path.insert(0, HERE)

from cases import FFT_IMAGES, SCORE_SETS, SEED  # noqa: E402
from oracle import texture_oracle as O  # noqa: E402
from sift_amd.synth import synth_clip, synth_frame  # noqa: E402

fft = {"numpy_version": np.__version__, "cases": []}
for idx, (kind, h, w) in enumerate(FFT_IMAGES):
    gray = O.np_gray(synth_frame(kind, SEED, 2, idx, h, w))
    mag = np.log(np.abs(np.fft.fftshift(np.fft.fft2(gray))) + 1)
    fft["cases"].append({"kind": kind, "h": h, "w": w, "frame": idx, "std": float(np.std(mag)),
                         "sum": float(np.sum(mag)), "sumsq": float(np.sum(mag * mag))})
with open(os.path.join(HERE, "fft_numpy.json"), "w") as f:
    json.dump(fft, f, indent=1)

# one set of goldens per gray coefficient set: 15-bit (the library's default here) and 14-bit
for shift, fname in ((15, "score_sets.json"), (14, "score_sets_gray14.json")):
    sets = {}
    for si, (name, kind, n, h, w) in enumerate(SCORE_SETS):
        frames = list(synth_clip(kind, SEED, 10 + si, n, h, w)) if n else []
        res = O.np_compute_texture_score(frames, shift=shift)
        ints = []
        for fr in frames[:10]:
            gray = O.np_gray(fr, shift)
            lbp = O.np_lbp_uniform(gray)
            hf = O.np_highpass(gray).astype(np.int64)
            blur = O.np_gauss5(gray)
            edges, _ = O.np_canny(gray)
            ints.append({
                "lbp_hist": np.bincount(lbp.ravel(), minlength=10).tolist(),
                "hp_sum": int(hf.sum()), "hp_sumsq": int((hf * hf).sum()),
                "absdiff_sum": int(np.abs(gray.astype(np.int64) - blur.astype(np.int64)).sum()),
                "edge_count": int((edges > 0).sum()), "gray_sum": int(gray.astype(np.int64).sum()),
            })
        sets[name] = {"kind": kind, "n": n, "h": h, "w": w, "video": 10 + si, "gray_shift": shift, "result": res,
                      "integers": ints}
    with open(os.path.join(HERE, fname), "w") as f:
        json.dump(sets, f, indent=1)
print("ok")
