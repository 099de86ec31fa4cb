"""Full-size checks on the GPU: BASELINE configs at their real shapes.

Where the CPU oracle is affordable (one 1080p frame ~2 s) the result is compared directly;
beyond that, size-independent properties are used: histograms sum to the pixel count, batch
results equal single-frame results, runs are bit-reproducible, device-resident and host-fed
paths agree, and frames generated on the device equal the host generator's.
"""
import ctypes as C

import numpy as np
import pytest

from cases import PATCHWORK_CASES, SEED
from patchwork import patchwork_gray
from oracle import texture_oracle as O
from sift_amd.synth import KIND_ID, synth_clip, synth_frame

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from sift_amd.texture import TextureContext
    c = TextureContext(0)
    yield c
    c.close()


def _ints(r):
    return (tuple(r.lbp_hist[:10]), r.edge_count, r.absdiff_sum, r.hp_sum, r.hp_sumsq, r.gray_sum, r.n_pixels)


def _floats(r):
    return tuple(getattr(r, k) for k in O.METRIC_KEYS) + (r.texture_score,)


@pytest.mark.parametrize("kind,h,w", [("natural", 1080, 1920), ("noise", 720, 1280), ("tie", 541, 963),
                                      ("step", 383, 1283),
                                      # the largest supported extent in each direction (STX_MAX_DIM)
                                      ("noise", 48, 8192), ("natural", 8192, 40)])
def test_full_frame_vs_oracle(ctx, coracle, kind, h, w):
    f = synth_frame(kind, SEED + 2, 1, 7, h, w)
    r = ctx.analyze_frames([f])[0]
    m = coracle.analyze_frame(f)
    assert _ints(r) == _ints(m)
    for a, b, k in zip(_floats(r), _floats(m), O.METRIC_KEYS + ("texture_score",)):
        assert a == pytest.approx(b, rel=1e-4, abs=1e-9), k
    gray, codes, edges = ctx.debug_planes(f)
    g = coracle.gray(f)
    assert np.array_equal(gray, g)
    assert np.array_equal(codes, coracle.lbp(g)[0])
    assert np.array_equal(edges, coracle.canny(g)[0])


@pytest.mark.parametrize("kind", ["ramp", "step", "edge"])
def test_1080p_structured_frequency_statistic(ctx, coracle, kind):
    """Noise-free structure at the headline frame size: ramp, bars with +-1 noise, a constant with one edge."""
    h, w = 1080, 1920
    if kind == "edge":
        g = np.full((h, w), 80, np.uint8)
        g[:, 1000:] = 140
        f = np.ascontiguousarray(np.repeat(g[:, :, None], 3, axis=2))
    else:
        f = synth_frame(kind, SEED + 2, 2, 3, h, w)
    r = ctx.analyze_frames([f])[0]
    m = coracle.analyze_frame(f)
    assert _ints(r) == _ints(m)
    for a, b, k in zip(_floats(r), _floats(m), O.METRIC_KEYS + ("texture_score",)):
        assert a == pytest.approx(b, rel=1e-4, abs=1e-9), k
    # larger than the LDS holds two fp64 pass buffers for: the second one goes through global memory
    if kind == "ramp":
        wide = np.ascontiguousarray(f[:6, :, :].repeat(4, axis=1)[:, :5120])  # 6 x 5120
        p = ctx.default_params(); p.fft_exact = 1
        want = float(np.std(np.log(np.abs(np.fft.fft2(coracle.gray(wide).astype(np.float64))) + 1)))
        assert ctx.analyze_frames([wide], p)[0].frequency_anomaly == pytest.approx(want, rel=1e-9)


def test_mixed_clip_at_full_size(ctx, coracle):
    """Several frames deep at the big shapes: every content class in one 1080p batch (one frame of each goes to the
    oracle), and three 4K frames of different classes through the default call (whole-frame FFT)."""
    h, w = 1080, 1920
    kinds = ("natural", "noise", "ramp", "flat", "tie", "step")
    frames = [synth_frame(k, SEED + 6, 3, i, h, w) for i, k in enumerate(kinds) for _ in range(2)]
    recs = ctx.analyze_frames(frames)
    assert len(recs) == len(frames)
    for i in range(0, len(frames), 2):
        m = coracle.analyze_frame(frames[i])
        assert _ints(recs[i]) == _ints(m), kinds[i // 2]
        for a, b, k in zip(_floats(recs[i]), _floats(m), O.METRIC_KEYS + ("texture_score",)):
            assert a == pytest.approx(b, rel=1e-4, abs=1e-9), (kinds[i // 2], k)
        # the same frame again in the batch: same record, whatever its neighbours are
        assert _ints(recs[i + 1]) == _ints(recs[i]) and _floats(recs[i + 1]) == _floats(recs[i])
    h, w = 2160, 3840
    frames = [synth_frame(k, SEED + 7, 4, i, h, w) for i, k in enumerate(("step", "natural", "tie"))]
    recs = ctx.analyze_frames(frames)
    for f, r, k in zip(frames, recs, ("step", "natural", "tie")):
        m = coracle.analyze_frame(f)
        assert _ints(r) == _ints(m), k
        for a, b, key in zip(_floats(r), _floats(m), O.METRIC_KEYS + ("texture_score",)):
            assert a == pytest.approx(b, rel=1e-4, abs=1e-9), (k, key)


@pytest.mark.parametrize("h,w,cell", PATCHWORK_CASES)
def test_flat_patches_vs_skimage_golden(ctx, h, w, cell):
    """Flat patches: every interior pixel is an exact tie of the four diagonal samples, resolved from the
    table for flat neighbourhoods.  Checked against scikit-image's own codes (histogram and digest)."""
    import hashlib
    import os
    g = patchwork_gray(h, w, cell)
    f = np.ascontiguousarray(np.repeat(g[:, :, None], 3, axis=2))  # B = G = R: the gray plane is g itself
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lbp_skimage.npz"))
    key = "patch_%dx%d_c%d" % (h, w, cell)
    r = ctx.analyze_frames([f])[0]
    assert np.array_equal(np.array(r.lbp_hist[:10], dtype=np.int64), gold[key + "_hist"])
    gray, codes, edges = ctx.debug_planes(f)
    assert np.array_equal(gray, g)
    assert hashlib.sha256(codes.tobytes()).hexdigest() == str(gold[key + "_sha256"])


def test_1080p_clip_properties(ctx, coracle):
    h, w, n = 1080, 1920, 24
    nbytes = n * h * w * 3
    dev = ctx.device_alloc(nbytes)
    try:
        ctx.synth_device(dev, KIND_ID["natural"], SEED + 2, 3, 0, n, h, w)
        res1 = ctx.analyze_device(dev, n, h, w)
        ints1 = [_ints(r) for r in res1[:n]]
        fl1 = [_floats(r) for r in res1[:n]]
        # histogram of codes covers every pixel; edge count bounded
        for r in res1[:n]:
            assert sum(r.lbp_hist[:10]) == h * w == r.n_pixels
            assert 0 < r.edge_count < h * w
            assert 0.0 <= r.texture_score <= 1.0
        # bit-reproducible, including the float statistics (fixed-order reductions)
        res2 = ctx.analyze_device(dev, n, h, w)
        assert [_ints(r) for r in res2[:n]] == ints1
        assert [_floats(r) for r in res2[:n]] == fl1
        # chunking does not change anything
        ctx.set_chunk_frames(5)
        res3 = ctx.analyze_device(dev, n, h, w)
        ctx.set_chunk_frames(0)
        assert [_ints(r) for r in res3[:n]] == ints1
        assert [_floats(r) for r in res3[:n]] == fl1
        # host-fed path on the same bytes
        host = ctx.d2h(dev, nbytes).reshape(n, h, w, 3)
        assert np.array_equal(host[5], synth_frame("natural", SEED + 2, 3, 5, h, w))
        res4 = ctx.analyze_batch(host[:6])
        assert [_ints(r) for r in res4[:6]] == ints1[:6]
        assert [_floats(r) for r in res4[:6]] == fl1[:6]
        # one frame of the clip against the oracle
        m = coracle.analyze_frame(host[11])
        assert _ints(res1[11]) == _ints(m)
        for a, b in zip(_floats(res1[11]), _floats(m)):
            assert a == pytest.approx(b, rel=1e-4, abs=1e-9)
    finally:
        ctx.device_free(dev)


def test_crop_batch_4096x256(ctx, coracle):
    """BASELINE configs[1]: 4,096 crops of 256x256."""
    n, h, w = 4096, 256, 256
    nbytes = n * h * w * 3
    dev = ctx.device_alloc(nbytes)
    try:
        for v in range(4):  # frame index < 1024 per video
            ctx.synth_device(dev + v * 1024 * h * w * 3, KIND_ID["natural"], SEED + 1, v, 0, 1024, h, w)
        res = ctx.analyze_device(dev, n, h, w)
        for idx in (0, 1, 1023, 1024, 2500, 4095):
            f = synth_frame("natural", SEED + 1, idx // 1024, idx % 1024, h, w)
            m = coracle.analyze_frame(f)
            assert _ints(res[idx]) == _ints(m), idx
            for a, b in zip(_floats(res[idx]), _floats(m)):
                assert a == pytest.approx(b, rel=1e-4, abs=1e-9)
        assert all(sum(r.lbp_hist[:10]) == h * w for r in res[:n])
    finally:
        ctx.device_free(dev)


def test_4k_multi_radius_and_fft_crop(ctx, coracle):
    """BASELINE configs[4] shape: 3840x2160, LBP at (8,1),(16,2),(24,3), FFT on the centred 512x512 crop."""
    h, w = 2160, 3840
    f = synth_frame("natural", SEED + 4, 0, 0, h, w)
    g = coracle.gray(f)
    for P, R in ((8, 1.0), (16, 2.0), (24, 3.0)):
        p = ctx.default_params()
        p.lbp_points, p.lbp_radius, p.fft_h, p.fft_w = P, R, 512, 512
        r = ctx.analyze_frames([f], p)[0]
        _, hist = coracle.lbp(g, P, R)
        assert list(r.lbp_hist[:P + 2]) == hist.tolist()
        if P == 8:
            op = coracle.default_params()
            op.fft_h = op.fft_w = 512
            m = coracle.analyze_frame(f, op)
            assert _ints(r) == _ints(m)
            for a, b in zip(_floats(r), _floats(m)):
                assert a == pytest.approx(b, rel=1e-4, abs=1e-9)
