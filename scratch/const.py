import sys
sys.path.insert(0, '.')
import numpy as np
from sift_amd.texture import TextureNoiseDetector
det = TextureNoiseDetector()
for (h, w) in ((480, 640), (1080, 1920)):
    for v in (0, 1, 77, 128, 200, 255):
        f = np.full((h, w, 3), v, np.uint8)
        m = det.analyze_frame(f)
        print(h, w, v, m["frequency_anomaly"], "numpy:", float(np.std(np.log(np.abs(np.fft.fft2(f[:, :, 0].astype(np.float64))) + 1))))
