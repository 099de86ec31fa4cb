import sys
sys.path.insert(0, '.')
import torch
from sift_amd.texture import TextureContext
from sift_amd._lib import StxMetrics
ctx = TextureContext(0)
ctx.set_concurrency(False)
for (h, w, nf) in ((256, 256, 4096), (1080, 1920, 150), (512, 512, 1024), (256, 1920, 640)):
    t = torch.empty(nf*h*w*3, dtype=torch.uint8, device='cuda')
    out = (StxMetrics*nf)()
    for label, kind in (("natural", 1), ("noise", 0)):
        for v0 in range(0, nf, 1024):
            n = min(1024, nf - v0)
            ctx.synth_device(t.data_ptr() + v0*h*w*3, kind, 0x5EED0002, v0 // 1024, 0, n, h, w)
        ctx.analyze_device(t.data_ptr(), nf, h, w, out=out)
        ctx.get_timing(reset=True); ctx.set_timing(True)
        for _ in range(3): ctx.analyze_device(t.data_ptr(), nf, h, w, out=out)
        ctx.set_timing(False)
        tm = ctx.get_timing(reset=True)["kernels"]
        px = nf*h*w
        print(h, w, nf, label, {k: round(v["ms"]/3, 3) for k, v in tm.items() if v["launches"]}, "stencil ps/px %.2f" % (tm["stencil"]["ms"]/3*1e9/px), flush=True)
    del t
