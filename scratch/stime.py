import sys
sys.path.insert(0, '.')
import torch
from sift_amd.texture import TextureContext
from sift_amd._lib import StxMetrics
ctx = TextureContext(0)
ctx.set_concurrency(False)
h, w, nf = 1080, 1920, 150
t = torch.empty(nf*h*w*3, dtype=torch.uint8, device='cuda')
out = (StxMetrics*nf)()
for label, kind in (("natural", 1), ("noise", 0), ("flat", 3), ("ramp", 2)):
    ctx.synth_device(t.data_ptr(), kind, 0x5EED0002, 0, 0, nf, h, w)
    ctx.analyze_device(t.data_ptr(), nf, h, w, out=out)
    ctx.get_timing(reset=True); ctx.set_timing(True)
    for _ in range(3): ctx.analyze_device(t.data_ptr(), nf, h, w, out=out)
    ctx.set_timing(False)
    tm = ctx.get_timing(reset=True)["kernels"]
    print(label, {k: round(v["ms"]/3, 3) for k, v in tm.items() if v["launches"]}, round(out[0].texture_score, 6), flush=True)
