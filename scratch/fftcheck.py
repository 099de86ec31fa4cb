import sys
sys.path.insert(0,'.')
import numpy as np
from sift_amd.texture import TextureContext
ctx = TextureContext(0)
sizes = [64,80,96,128,160,240,256,320,360,480,512,540,640,720,768,960,1024,1080,1280,1440,1920,2048]
rng = np.random.default_rng(0)
def ref(gray):
    mag = np.log(np.abs(np.fft.fft2(gray)) + 1); return float(np.std(mag))
for n in sizes:
    for (h,w) in ((n, 66), (66, n)):   # isolate rows (w=n) and cols (h=n); 66 is a generic size
        g = rng.integers(0,256,(h,w),dtype=np.uint8)
        f = np.ascontiguousarray(np.repeat(g[...,None],3,axis=2))
        r = ctx.analyze_frames([f])[0]
        want = ref(g)
        rel = abs(r.frequency_anomaly-want)/want
        print(n, (h,w), 'rel %.2e' % rel, 'BAD' if rel>1e-5 else '')
