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
def run(label, kind, **kw):
    ctx.synth_device(t.data_ptr(), kind, 0x5EED0002, 0, 0, nf, h, w)
    p = ctx.default_params()
    for k, v in kw.items(): setattr(p, k, v)
    ctx.analyze_device(t.data_ptr(), nf, h, w, params=p, out=out)
    ctx.get_timing(reset=True); ctx.set_timing(True)
    for _ in range(3): ctx.analyze_device(t.data_ptr(), nf, h, w, params=p, out=out)
    ctx.set_timing(False)
    tm = ctx.get_timing(reset=True)["kernels"]
    print(label, {k: round(v["ms"]/3, 3) for k, v in tm.items() if v["launches"]}, "edge", round(out[0].edge_density, 4), flush=True)
run("natural full      ", 1)
run("natural P8 R2      ", 1, lbp_radius=2.0); run("natural P16 R2     ", 1, lbp_points=16, lbp_radius=2.0); run("natural P24 R3     ", 1, lbp_points=24, lbp_radius=3.0); run("noise P24 R3       ", 0, lbp_points=24, lbp_radius=3.0)
run("natural noNMS     ", 1, canny_low=30000, canny_high=30001)
run("natural noLBP noNMS", 1, lbp_radius=2.0, canny_low=30000, canny_high=30001)
run("noise full        ", 0)
run("flat full         ", 3)
run("ramp full         ", 2)
