"""Two extended calls on a few 4K frames (BASELINE configs[4]: three LBP radii, 512x512 FFT crop, Sobel / blockiness
sums), for a rocprofv3 counter pass."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sift_amd.texture import TextureContext
ctx = TextureContext(0)
h, w, nf = 2160, 3840, 10
t = torch.empty(nf * h * w * 3, dtype=torch.uint8, device='cuda')
ctx.synth_device(t.data_ptr(), 1, 0x5EED0004, 0, 0, nf, h, w)
p = ctx.default_params()
p.fft_h = p.fft_w = 512
for i in range(2):
    m, x = ctx.analyze_device_extended(t.data_ptr(), nf, h, w, p)
print(m[0].texture_score, list(x[0].lbp_hist_p24r3[:4]))
