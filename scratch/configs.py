import sys, time
sys.path.insert(0, '.')
import torch
from sift_amd.texture import TextureContext
from sift_amd._lib import StxMetrics
ctx = TextureContext(0)
def run(label, h, w, nf, **kw):
    t = torch.empty(nf*h*w*3, dtype=torch.uint8, device='cuda')
    out = (StxMetrics*nf)()
    for v0 in range(0, nf, 1024):
        n = min(1024, nf - v0)
        ctx.synth_device(t.data_ptr() + v0*h*w*3, 1, 0x5EED0001, v0 // 1024, 0, n, h, w)
    p = ctx.default_params()
    for k, v in kw.items(): setattr(p, k, v)
    ctx.analyze_device(t.data_ptr(), nf, h, w, params=p, out=out)
    ctx.get_timing(reset=True); ctx.set_timing(True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): ctx.analyze_device(t.data_ptr(), nf, h, w, params=p, out=out)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
    ctx.set_timing(False)
    tm = ctx.get_timing(reset=True)["kernels"]
    print(label, "%.1f frames/s  %.3f ms" % (nf / dt, dt * 1e3), {k: round(v["ms"]/3, 3) for k, v in tm.items() if v["launches"]}, flush=True)
    del t
run("cfg2 4096 x 256x256      ", 256, 256, 4096)
run("cfg1 150 x 720p          ", 720, 1280, 150)
run("cfg1 150 x 80x80         ", 80, 80, 150)
run("cfg5 4K P8R1 fft512  x40 ", 2160, 3840, 40, fft_h=512, fft_w=512)
run("cfg5 4K P16R2 fft512 x40 ", 2160, 3840, 40, lbp_points=16, lbp_radius=2.0, fft_h=512, fft_w=512)
run("cfg5 4K P24R3 fft512 x40 ", 2160, 3840, 40, lbp_points=24, lbp_radius=3.0, fft_h=512, fft_w=512)
def run_ext(label, h, w, nf, **kw):
    t = torch.empty(nf*h*w*3, dtype=torch.uint8, device='cuda')
    ctx.synth_device(t.data_ptr(), 1, 0x5EED0001, 0, 0, nf, h, w)
    p = ctx.default_params()
    for k, v in kw.items(): setattr(p, k, v)
    ctx.analyze_device_extended(t.data_ptr(), nf, h, w, params=p)
    ctx.get_timing(reset=True); ctx.set_timing(True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): ctx.analyze_device_extended(t.data_ptr(), nf, h, w, params=p)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
    ctx.set_timing(False)
    tm = ctx.get_timing(reset=True)["kernels"]
    print(label, "%.1f frames/s  %.3f ms" % (nf / dt, dt * 1e3), {k: round(v["ms"]/3, 3) for k, v in tm.items() if v["launches"]}, flush=True)
    del t
run_ext("cfg5 4K all three radii + sobel + blockiness, fft512 x40 ", 2160, 3840, 40, fft_h=512, fft_w=512)
run_ext("1080p extended x150 ", 1080, 1920, 150)
