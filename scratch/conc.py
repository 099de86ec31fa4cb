import sys, time
sys.path.insert(0, '.')
import torch
from sift_amd.texture import TextureContext
from sift_amd._lib import StxMetrics
ctx = TextureContext(0)
for (h, w, nf) in ((256, 256, 4096), (1080, 1920, 150), (720, 1280, 150), (80, 80, 150)):
    t = torch.empty(nf*h*w*3, dtype=torch.uint8, device='cuda')
    out = (StxMetrics*nf)()
    for v0 in range(0, nf, 1024):
        n = min(1024, nf - v0)
        ctx.synth_device(t.data_ptr() + v0*h*w*3, 1, 0x5EED0002, v0 // 1024, 0, n, h, w)
    for conc in (False, True, False, True):
        ctx.set_concurrency(conc)
        ctx.analyze_device(t.data_ptr(), nf, h, w, out=out)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): ctx.analyze_device(t.data_ptr(), nf, h, w, out=out)
        wall = (time.perf_counter() - t0) / 5 * 1e3
        print(h, w, nf, "conc", conc, "wall %.3f ms" % wall, flush=True)
    del t
