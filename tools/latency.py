import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sift_amd.texture import TextureNoiseDetector
from sift_amd.synth import synth_frame
det = TextureNoiseDetector()
for (h, w) in ((480, 640), (720, 1280), (1080, 1920)):
    frames = [synth_frame("natural", 0x5EED0000, 0, i, h, w) for i in range(10)]
    det.compute_texture_score(frames)
    ts = []
    for _ in range(20):
        t0 = time.perf_counter(); r = det.compute_texture_score(frames); ts.append(time.perf_counter() - t0)
    t1 = []
    for _ in range(20):
        t0 = time.perf_counter(); det.analyze_frame(frames[0]); t1.append(time.perf_counter() - t0)
    print(h, w, "compute_texture_score(10 frames): median %.2f ms; analyze_frame: median %.2f ms" % (np.median(ts) * 1e3, np.median(t1) * 1e3), r["average_score"], flush=True)
# odd width and strided views
for (h, w) in ((481, 642), (480, 641)):
    frames = [synth_frame("natural", 0x5EED0000, 0, i, h, w) for i in range(10)]
    det.compute_texture_score(frames)
    ts = []
    for _ in range(10):
        t0 = time.perf_counter(); r = det.compute_texture_score(frames); ts.append(time.perf_counter() - t0)
    print(h, w, "odd: compute_texture_score(10 frames): median %.2f ms" % (np.median(ts) * 1e3), flush=True)
big = [synth_frame("natural", 0x5EED0000, 0, i, 480, 800) for i in range(10)]
views = [b[:, 80:720] for b in big]
det.compute_texture_score(views)
ts = []
for _ in range(10):
    t0 = time.perf_counter(); r = det.compute_texture_score(views); ts.append(time.perf_counter() - t0)
print("views: median %.2f ms" % (np.median(ts) * 1e3), flush=True)
