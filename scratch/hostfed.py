import sys, time
sys.path.insert(0,'.')
import numpy as np, torch
from sift_amd.texture import TextureContext
ctx = TextureContext(0)
h,w,nf=1080,1920,150
t = torch.empty(nf*h*w*3, dtype=torch.uint8, device='cuda')
ctx.synth_device(t.data_ptr(), 1, 0x5EED0002, 0, 0, nf, h, w)
host = ctx.d2h(t.data_ptr(), nf*h*w*3).reshape(nf,h,w,3)
pinned = torch.empty((nf,h,w,3), dtype=torch.uint8).pin_memory()
pinned.numpy()[...] = host
for label, arr in (('pageable', host), ('pinned', pinned.numpy())):
    for chunk in (0, 30):
        ctx.set_chunk_frames(chunk)
        ctx.analyze_batch(arr)
        t0=time.perf_counter()
        for i in range(3): ctx.analyze_batch(arr)
        dt=(time.perf_counter()-t0)/3
        print(label, 'chunk', chunk, 'ms/clip %.2f' % (dt*1e3), 'frames/s %.0f' % (nf/dt), 'GB/s %.1f' % (nf*h*w*3/dt/1e9))
