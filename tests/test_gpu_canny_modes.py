"""Where Canny's non-maximum suppression runs must not change anything: the suppression on the pixels a flood of the
raw threshold maps reaches ("sparse"), the suppression in the stencil pass ("dense") and the adaptive default give
the records of the CPU oracle (cv2.Canny(gray, 50, 150) of /root/reference/app/passive_checker.py:116, edge count
bit-exact), including the frames where the sparse path gives up and the chunk goes through again the dense way.
"""
import ctypes as C

import numpy as np
import pytest

from cases import SEED
from sift_amd.synth import KINDS, synth_frame

pytestmark = pytest.mark.gpu

INTS = ("edge_count", "absdiff_sum", "hp_sum", "hp_sumsq", "gray_sum")


@pytest.fixture(scope="module")
def ctxs():
    from sift_amd.texture import TextureContext
    out = {}
    for mode in ("dense", "sparse", "auto"):
        c = TextureContext(0)
        c.set_canny_mode(mode)
        out[mode] = c
    yield out
    for c in out.values():
        c.close()


def _ints(r):
    return tuple(int(getattr(r, k)) for k in INTS) + tuple(r.lbp_hist[:10])


@pytest.mark.parametrize("kind", KINDS)
def test_modes_agree_with_each_other_and_the_oracle(ctxs, coracle, kind):
    sizes = [(1, 1), (2, 3), (3, 8), (5, 64), (64, 64), (65, 129), (130, 131), (257, 300), (200, 516), (720, 1280)]
    for h, w in sizes:
        frames = [synth_frame(kind, SEED, 11, i, h, w) for i in range(2)]
        want = [coracle.canny(coracle.gray(f))[2] for f in frames]
        recs = {m: c.analyze_frames(frames) for m, c in ctxs.items()}
        for i in range(len(frames)):
            assert recs["dense"][i].edge_count == want[i], (kind, h, w)
            for m in ("sparse", "auto"):
                assert _ints(recs[m][i]) == _ints(recs["dense"][i]), (kind, h, w, m)
                assert recs[m][i].frequency_anomaly == recs["dense"][i].frequency_anomaly


def test_edges_on_the_image_border_and_threshold_knobs(ctxs, coracle):
    # bright frame lines on the first / last rows and columns, a diagonal, and blocks that touch the corners
    h, w = 150, 260
    g = np.full((h, w), 30, np.uint8)
    g[0, :] = 200
    g[-1, :] = 220
    g[:, 0] = 210
    g[:, -1] = 190
    for k in range(10, 120):
        g[k, k + 20] = 255
    g[:6, :6] = 255
    g[-5:, -7:] = 0
    rng = np.random.default_rng(5)
    g[40:110, 150:240] = rng.integers(0, 256, (70, 90), dtype=np.uint8)
    bgr = np.ascontiguousarray(np.repeat(g[..., None], 3, axis=2))
    for low, high in ((50, 150), (0, 0), (10, 2000), (150, 50), (0, 5000), (2039, 2040)):
        got = {}
        for m, c in ctxs.items():
            p = c.default_params()
            p.canny_low, p.canny_high = low, high
            got[m] = c.analyze_frames([bgr], p)[0].edge_count
        assert got["sparse"] == got["dense"] == got["auto"], (low, high, got)
    assert ctxs["sparse"].analyze_frames([bgr])[0].edge_count == coracle.canny(g)[2]


def test_long_contours_fall_back_to_the_dense_path(ctxs, coracle):
    # one weak path hanging off one seed, crossing the frame back and forth: the flood of the raw maps is still
    # growing when its passes are used up, so the sparse path hands the chunk back
    h, w = 520, 700
    g = np.full((h, w), 40, np.uint8)
    ys = list(range(8, h - 8, 10))
    for k, y in enumerate(ys):
        g[y:y + 3, 8:w - 8] = 60
        if k + 1 < len(ys):
            x = w - 11 if k % 2 == 0 else 8
            g[y:ys[k + 1] + 3, x:x + 3] = 60
    g[6:12, 8:14] = 255
    snake = np.ascontiguousarray(np.repeat(g[..., None], 3, axis=2))
    frames = [snake, synth_frame("noise", SEED, 5, 0, h, w), synth_frame("natural", SEED, 5, 1, h, w)]
    want = [coracle.canny(coracle.gray(f))[2] for f in frames]
    for m, c in ctxs.items():
        for _ in range(2):  # (the adaptive context changes its mind between the two calls)
            assert [r.edge_count for r in c.analyze_frames(frames)] == want, m


def test_dense_content_overflows_the_list_and_falls_back(ctxs, coracle):
    # every pixel of a noise frame is reached by the flood: far beyond the 4 % a frame's list holds
    frames = [synth_frame("noise", SEED, 6, i, 300, 420) for i in range(3)] + [synth_frame("natural", SEED, 6, 3, 300, 420)]
    want = [coracle.canny(coracle.gray(f))[2] for f in frames]
    for m, c in ctxs.items():
        assert [r.edge_count for r in c.analyze_frames(frames)] == want, m


def test_device_clip_and_async_submission_in_sparse_mode(ctxs):
    import torch
    from sift_amd._lib import StxMetrics
    from sift_amd.synth import KIND_ID
    nf, h, w = 24, 360, 640
    t = torch.empty(nf * h * w * 3, dtype=torch.uint8, device="cuda")
    ctxs["dense"].synth_device(t.data_ptr(), KIND_ID["natural"], SEED, 3, 0, nf, h, w)
    torch.cuda.synchronize()
    outs = {}
    for m, c in ctxs.items():
        out = (StxMetrics * nf)()
        c.analyze_device(t.data_ptr(), nf, h, w, out=out)
        outs[m] = bytes(C.string_at(C.addressof(out), C.sizeof(out)))
        out2 = (StxMetrics * nf)()
        c.wait(c.submit_device(t.data_ptr(), nf, h, w, out2))
        assert bytes(C.string_at(C.addressof(out2), C.sizeof(out2))) == outs[m], m
    assert outs["sparse"] == outs["dense"] == outs["auto"]
