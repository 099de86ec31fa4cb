import sys, time, os
sys.path.insert(0, os.getcwd())
import numpy as np
from sift_amd.texture import TextureContext
from sift_amd.synth import synth_frame
ctx = TextureContext(0)
for (h, w) in ((4099, 4099), (1031, 1031), (521, 523), (1080, 1920)):
    f = synth_frame("noise", 1, 0, 0, h, w)
    ctx.analyze_frames([f])
    t = time.perf_counter()
    for _ in range(3): r = ctx.analyze_frames([f])[0]
    dt = (time.perf_counter() - t) / 3
    ref = float(np.std(np.log(np.abs(np.fft.fft2(ctx.debug_planes(f)[0])) + 1)))
    print(h, w, "ms/frame (host-fed, one frame)", round(dt * 1e3, 2), "rel err", abs(r.frequency_anomaly - ref) / ref)
