import sys, time
sys.path.insert(0, '.')
import numpy as np
from sift_amd.texture import TextureNoiseDetector
from sift_amd.synth import synth_frame
det = TextureNoiseDetector()
sizes = [(1, 1), (7, 5), (16, 16), (80, 80), (137, 151), (211, 199), (256, 256), (333, 517), (500, 499), (1000, 1000), (1023, 1025),
         (1080, 1920), (1200, 1600), (1440, 2560), (2160, 3840), (3000, 4000), (4099, 4099), (8192, 100), (100, 8192), (8192, 8192)]
for (h, w) in sizes:
    for kind in ("natural", "flat"):
        f = synth_frame(kind, 0x5EED0000, 0, 0, h, w)
        det.analyze_frame(f)
        t0 = time.perf_counter(); n = 3
        for _ in range(n): det.analyze_frame(f)
        dt = (time.perf_counter() - t0) / n
        print("%5d x %5d %-8s %9.3f ms  %7.2f ns/px" % (h, w, kind, dt * 1e3, dt * 1e9 / (h * w)), flush=True)
