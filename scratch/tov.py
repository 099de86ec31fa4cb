import sys, time
sys.path.insert(0, '.')
import torch
from sift_amd.texture import TextureContext
from sift_amd._lib import StxMetrics
ctx = TextureContext(0)
h, w, nf = 1080, 1920, 150
t = torch.empty(nf*h*w*3, dtype=torch.uint8, device='cuda')
out = (StxMetrics*nf)()
ctx.synth_device(t.data_ptr(), 1, 0x5EED0002, 0, 0, nf, h, w)
for timing in (False, True, False, True):
    ctx.set_timing(timing)
    ctx.analyze_device(t.data_ptr(), nf, h, w, out=out)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(8): ctx.analyze_device(t.data_ptr(), nf, h, w, out=out)
    wall = (time.perf_counter() - t0) / 8 * 1e3
    print("timing", timing, "wall %.3f ms" % wall, flush=True)
