import sys, json
sys.path.insert(0,'.')
import torch
from sift_amd.texture import TextureContext
from sift_amd._lib import StxMetrics
ctx = TextureContext(0)
h,w,nf=1080,1920,150
t = torch.empty(nf*h*w*3, dtype=torch.uint8, device='cuda')
ctx.synth_device(t.data_ptr(), 1, 0x5EED0002, 0, 0, nf, h, w)
out=(StxMetrics*nf)()
for label, P in (('lbp81',8),('lbp16',16)):
    p = ctx.default_params(); p.lbp_points=P; p.lbp_radius = 1.0 if P==8 else 2.0
    ctx.analyze_device(t.data_ptr(), nf, h, w, params=p, out=out)
    ctx.get_timing(reset=True); ctx.set_timing(True)
    for i in range(3): ctx.analyze_device(t.data_ptr(), nf, h, w, params=p, out=out)
    ctx.set_timing(False)
    tm = ctx.get_timing()
    print(label, {k: round(v['ms']/3,3) for k,v in tm['kernels'].items() if v['launches']})
