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
    for chunk in (0, 75, 50, 25, 15):
        ctx.set_chunk_frames(chunk)
        ctx.analyze_device(t.data_ptr(), nf, h, w, out=out)
        ctx.get_timing(reset=True); ctx.set_timing(not conc)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): ctx.analyze_device(t.data_ptr(), nf, h, w, out=out)
        wall = (time.perf_counter() - t0) / 5 * 1e3
        ctx.set_timing(False)
        tm = ctx.get_timing(reset=True)["kernels"]
        print("conc", conc, "chunk", chunk, "wall %.3f" % wall, {k: round(v["ms"]/5, 3) for k, v in tm.items() if v["launches"]}, flush=True)
