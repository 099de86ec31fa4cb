import sys, time
sys.path.insert(0, '.')
import numpy as np
from sift_amd.checkers import FallbackPassiveChecker, ImprovedFallbackPassiveChecker
from sift_amd.preprocess import FramePreprocessor
from sift_amd.extended import ExtendedTextureAnalyzer
from sift_amd.frames import PinnedFrameBatch
from sift_amd.synth import synth_frame
h, w = 480, 640
frames = [synth_frame("natural", 0x5EED0000, 0, i, h, w) for i in range(15)]
def med(f, n=15):
    f(); ts = []
    for _ in range(n):
        t0 = time.perf_counter(); f(); ts.append(time.perf_counter() - t0)
    return np.median(ts) * 1e3
a = FallbackPassiveChecker(); b = ImprovedFallbackPassiveChecker(); p = FramePreprocessor(); e = ExtendedTextureAnalyzer()
print("FallbackPassiveChecker.check(15 x 480p)         %.2f ms" % med(lambda: a.check(frames)))
print("ImprovedFallbackPassiveChecker.check(15 x 480p) %.2f ms" % med(lambda: b.check(frames)))
print("FramePreprocessor.preprocess_frames(15 x 480p)  %.2f ms" % med(lambda: p.preprocess_frames(frames)))
print("FramePreprocessor._preprocess_frame(480p)       %.2f ms" % med(lambda: p._preprocess_frame(frames[0])))
print("ExtendedTextureAnalyzer.analyze_frames(15)      %.2f ms" % med(lambda: e.analyze_frames(frames)))
batch = PinnedFrameBatch(15, h, w)
for f in frames: batch.append(f)
print("PinnedFrameBatch.analyze(15 x 480p)             %.2f ms" % med(lambda: batch.analyze()))
