import sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sift_amd.texture import TextureContext
from sift_amd._lib import StxMetrics
ctx = TextureContext(0)
# the path the benchmark's 150-frame clips take (auto goes sparse from 64 Mpixel per dispatch; these are 30 frames)
ctx.set_canny_mode(os.environ.get('STX_PROF_CANNY', 'sparse'))
h,w,nf=1080,1920,30
t = torch.empty(nf*h*w*3, dtype=torch.uint8, device='cuda')
ctx.synth_device(t.data_ptr(), 1, 0x5EED0002, 0, 0, nf, h, w)
out=(StxMetrics*nf)()
for i in range(2): ctx.analyze_device(t.data_ptr(), nf, h, w, out=out)
print(out[0].texture_score)
