import sys, time
sys.path.insert(0,'.')
import torch
from sift_amd.texture import TextureContext
from sift_amd._lib import StxMetrics
ctx = TextureContext(0)
h,w,nf=1080,1920,150
t = torch.empty(nf*h*w*3, dtype=torch.uint8, device='cuda')
ctx.synth_device(t.data_ptr(), 1, 0x5EED0002, 0, 0, nf, h, w)
out=(StxMetrics*nf)()
for chunk in (0, 5, 10, 15, 25, 30, 50, 75, 150):
    ctx.set_chunk_frames(chunk)
    ctx.analyze_device(t.data_ptr(), nf, h, w, out=out)
    ctx.get_timing(reset=True); ctx.set_timing(True)
    torch.cuda.synchronize(); t0=time.perf_counter()
    for i in range(4): ctx.analyze_device(t.data_ptr(), nf, h, w, out=out)
    torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/4
    ctx.set_timing(False)
    tm = ctx.get_timing()
    ks = {k: round(v['ms']/4,3) for k,v in tm['kernels'].items() if v['launches']}
    print('chunk', chunk, 'ms/step %.3f' % (dt*1e3), 'kernels sum %.3f' % sum(ks.values()), ks)
