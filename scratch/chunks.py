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
for conc in (False, True):
    ctx.set_concurrency(conc)
    for cf in (150, 75, 50, 30, 15, 10):
        ctx.set_chunk_frames(cf)
        ctx.analyze_device(t.data_ptr(), nf, h, w, out=out)
        ctx.get_timing(reset=True); ctx.set_timing(True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(4): ctx.analyze_device(t.data_ptr(), nf, h, w, out=out)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 4 * 1e3
        ctx.set_timing(False)
        tm = ctx.get_timing(reset=True)["kernels"]
        print("conc", conc, "chunk", cf, "wall %.3f" % dt, {k: round(v["ms"]/4, 3) for k, v in tm.items() if v["launches"]}, flush=True)
