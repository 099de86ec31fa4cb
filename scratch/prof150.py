import sys
sys.path.insert(0,'.')
import torch
from sift_amd.texture import TextureContext
from sift_amd._lib import StxMetrics
kind = int(sys.argv[1]) if len(sys.argv) > 1 else 1
ctx = TextureContext(0)
ctx.set_concurrency(0)
h,w,nf=1080,1920,150
t = torch.empty(nf*h*w*3, dtype=torch.uint8, device='cuda')
ctx.synth_device(t.data_ptr(), kind, 0x5EED0002, 0, 0, nf, h, w)
out=(StxMetrics*nf)()
for i in range(3): ctx.analyze_device(t.data_ptr(), nf, h, w, out=out)
print(out[0].texture_score)
