import sys, time
sys.path.insert(0, '.')
import numpy as np
from sift_amd.texture import TextureNoiseDetector
from sift_amd.synth import synth_frame
det = TextureNoiseDetector()
h, w = 480, 640
frames = [synth_frame("natural", 0x5EED0000, 0, i, h, w) for i in range(10)]
det.analyze_frame(frames[0])
t0 = time.perf_counter()
for _ in range(50): det.analyze_frame(frames[0])
print("analyze_frame %.3f ms" % ((time.perf_counter() - t0) / 50 * 1e3))
t0 = time.perf_counter()
for _ in range(50): det._ctx.analyze_frames(frames)
print("ctx.analyze_frames(10) %.3f ms" % ((time.perf_counter() - t0) / 50 * 1e3))
