"""GPU parity: the HIP path, called through the C ABI, against the CPU oracle.

Integer quantities (gray plane, LBP codes and histogram, Canny edge map and count, high-pass
sum / sum of squares, |gray-blur| sum) must be bit-exact; float metrics and scores must agree
within 1e-4 relative (the FFT runs in fp32 on the device, fp64 in numpy).
"""
import json
import os

import numpy as np
import pytest

from cases import LBP_IMAGES, SCORE_SETS, SEED
from oracle import texture_oracle as O
from sift_amd.synth import KIND_ID, KINDS, synth_clip, synth_frame

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden")
REL = 1e-4


@pytest.fixture(scope="module")
def ctx():
    from sift_amd.texture import TextureContext
    c = TextureContext(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def det():
    from sift_amd.texture import TextureNoiseDetector
    return TextureNoiseDetector(0)


def _check_record(r, m, P=8, rel=REL):
    assert list(r.lbp_hist[:P + 2]) == list(m.lbp_hist[:P + 2])
    assert r.edge_count == m.edge_count
    assert r.absdiff_sum == m.absdiff_sum
    assert r.hp_sum == m.hp_sum and r.hp_sumsq == m.hp_sumsq
    assert r.gray_sum == m.gray_sum and r.n_pixels == m.n_pixels
    for k in O.METRIC_KEYS:
        a, b = getattr(r, k), getattr(m, k)
        assert a == pytest.approx(b, rel=rel, abs=1e-9), k
    assert r.texture_score == pytest.approx(m.texture_score, rel=rel, abs=1e-9)


SIZES = [(1, 1), (1, 7), (5, 1), (2, 2), (3, 3), (5, 7), (24, 31), (33, 65), (64, 64), (45, 77), (80, 80),
         (100, 130), (128, 192), (67, 200),
         # around the 128-pixel stencil tiles, the 32-row strips and the 64-row bitmap words
         (129, 129), (130, 257), (257, 131), (97, 515), (300, 390)]


@pytest.mark.parametrize("kind", KINDS)
def test_planes_bit_exact(ctx, coracle, kind):
    for (h, w) in SIZES:
        f = synth_frame(kind, SEED, 3, 1, h, w)
        gray, codes, edges = ctx.debug_planes(f)
        g = coracle.gray(f)
        assert np.array_equal(gray, g), ("gray", kind, h, w)
        want_codes, _ = coracle.lbp(g, 8, 1.0)
        assert np.array_equal(codes, want_codes), ("lbp", kind, h, w, int((codes != want_codes).sum()))
        want_edges, _, _ = coracle.canny(g)
        assert np.array_equal(edges, want_edges), ("canny", kind, h, w, int((edges != want_edges).sum()))


@pytest.mark.parametrize("kind", KINDS)
def test_metrics_vs_oracle(ctx, coracle, kind):
    for (h, w) in SIZES:
        f = synth_frame(kind, SEED, 4, 2, h, w)
        r = ctx.analyze_frames([f])[0]
        m = coracle.analyze_frame(f)
        _check_record(r, m)


def test_lbp_codes_vs_skimage_golden(ctx):
    gold = np.load(os.path.join(GOLD, "lbp_skimage.npz"))
    p16 = ctx.default_params(); p16.lbp_points = 16; p16.lbp_radius = 2.0
    p24 = ctx.default_params(); p24.lbp_points = 24; p24.lbp_radius = 3.0
    for idx, (kind, h, w) in enumerate(LBP_IMAGES):
        g = synth_frame(kind, SEED, 0, idx, h, w)[..., 1]
        bgr = np.ascontiguousarray(np.repeat(g[..., None], 3, axis=2))  # gray(v,v,v) == v
        for P, prm in ((8, None), (16, p16), (24, p24)):
            gray, codes, _ = ctx.debug_planes(bgr, prm)
            assert np.array_equal(gray, g)
            want = gold["lbp_%02d_%s_%dx%d_P%d" % (idx, kind, h, w, P)]
            assert np.array_equal(codes, want), (kind, h, w, P, int((codes != want).sum()))


def test_lbp_near_tie_lattice_vs_skimage(ctx, coracle):
    """The adversarial image of the goldens: many diagonal samples a few 1e-4 gray levels off the centre, i.e.
    just inside and just outside the error margin of the fixed-point pre-decision."""
    from lattice import lattice_image
    gold = np.load(os.path.join(GOLD, "lbp_skimage.npz"))
    g = lattice_image()
    p16 = ctx.default_params(); p16.lbp_points = 16; p16.lbp_radius = 2.0
    p24 = ctx.default_params(); p24.lbp_points = 24; p24.lbp_radius = 3.0
    settings = ((8, 1.0, None), (16, 2.0, p16), (24, 3.0, p24))
    bgr = np.ascontiguousarray(np.repeat(g[..., None], 3, axis=2))
    for P, R, prm in settings:
        _, codes, _ = ctx.debug_planes(bgr, prm)
        want = gold["lbp_lattice_P%d" % P]
        assert np.array_equal(codes, want), (P, int((codes != want).sum()))
    # Embedded at several offsets, so that the same neighbourhoods meet interior and border strips, both pixels
    # of a lane and every row of a block.  The outcome of an exact tie depends on the pixel's coordinates (the
    # fp64 weights do), so these are checked against the C oracle, itself pinned by the golden above.
    for (oy, ox) in ((1, 1), (37, 130), (2, 127)):
        big = np.full((g.shape[0] + oy + 3, g.shape[1] + ox + 5), 100, np.uint8)
        big[oy:oy + g.shape[0], ox:ox + g.shape[1]] = g
        bgr = np.ascontiguousarray(np.repeat(big[..., None], 3, axis=2))
        for P, R, prm in settings:
            _, codes, _ = ctx.debug_planes(bgr, prm)
            want, _ = coracle.lbp(big, P, R)
            assert np.array_equal(codes, want), (oy, ox, P, int((codes != want).sum()))


@pytest.mark.parametrize("shift,fname", [(15, "score_sets.json"), (14, "score_sets_gray14.json")])
def test_score_sets_golden(ctx, shift, fname):
    from sift_amd.texture import TextureNoiseDetector
    gold = json.load(open(os.path.join(GOLD, fname)))
    prm = ctx.default_params()
    prm.gray_shift = shift
    det = TextureNoiseDetector(device=0, params=prm)
    for si, (name, kind, n, h, w) in enumerate(SCORE_SETS):
        entry = gold[name]
        frames = list(synth_clip(kind, SEED, 10 + si, n, h, w)) if n else []
        res = det.compute_texture_score(frames)
        want = entry["result"]
        assert list(res.keys()) == ["passed", "average_score", "individual_scores", "detailed_metrics", "message"]
        assert res["passed"] is want["passed"]
        assert res["average_score"] == pytest.approx(want["average_score"], abs=1.001e-3)
        assert len(res["individual_scores"]) == min(n, 10)
        for a, b in zip(res["individual_scores"], want["individual_scores"]):
            assert a == pytest.approx(b, abs=1.001e-3)
        for d, wd in zip(res["detailed_metrics"], want["detailed_metrics"]):
            assert list(d.keys()) == list(O.METRIC_KEYS)
            for k in O.METRIC_KEYS:
                assert isinstance(d[k], float)
                assert d[k] == pytest.approx(wd[k], rel=REL, abs=1e-9), (name, k)
        recs = det.analyze_frames_raw(frames[:10])
        for r, ints in zip(recs, entry["integers"]):
            assert list(r.lbp_hist[:10]) == ints["lbp_hist"]
            assert (r.hp_sum, r.hp_sumsq, r.absdiff_sum, r.edge_count, r.gray_sum) == (
                ints["hp_sum"], ints["hp_sumsq"], ints["absdiff_sum"], ints["edge_count"], ints["gray_sum"])
        json.dumps(res)  # JSON-native leaves


def test_batch_chunking_and_strides(ctx, coracle):
    clip = synth_clip("natural", SEED, 21, 7, 72, 136)
    want = [coracle.analyze_frame(f) for f in clip]
    for chunk in (0, 1, 3):
        ctx.set_chunk_frames(chunk)
        res = ctx.analyze_batch(clip)
        for r, m in zip(res, want):
            _check_record(r, m)
    ctx.set_chunk_frames(0)
    # non-contiguous views: every second column of a wider clip, and a row-padded buffer
    wide = synth_clip("noise", SEED, 22, 2, 40, 90)
    views = [wide[i][:, ::2, :] for i in range(2)]
    res = ctx.analyze_frames(views)
    for r, v in zip(res, views):
        _check_record(r, coracle.analyze_frame(np.ascontiguousarray(v)))
    padded = np.zeros((40, 400), np.uint8)
    fr = synth_frame("natural", SEED, 23, 0, 40, 121)
    padded[:, :363] = fr.reshape(40, 363)
    view = np.lib.stride_tricks.as_strided(padded, shape=(40, 121, 3), strides=(400, 3, 1))
    _check_record(ctx.analyze_frames([view])[0], coracle.analyze_frame(fr))


def test_device_resident_and_synth(ctx, coracle):
    h, w, n = 96, 160, 5
    nbytes = n * h * w * 3
    dev = ctx.device_alloc(nbytes)
    try:
        for kind in KINDS:
            ctx.synth_device(dev, KIND_ID[kind], SEED + 2, 5, 3, n, h, w)
            got = ctx.d2h(dev, nbytes).reshape(n, h, w, 3)
            want = synth_clip(kind, SEED + 2, 5, n, h, w, frame0=3)
            assert np.array_equal(got, want), kind
            res = ctx.analyze_device(dev, n, h, w)
            for i in range(n):
                _check_record(res[i], coracle.analyze_frame(want[i]))
    finally:
        ctx.device_free(dev)


def test_params_fft_crop_gray14_thresholds(ctx, coracle):
    f = synth_frame("natural", SEED, 30, 0, 120, 168)
    p = ctx.default_params()
    p.fft_h, p.fft_w = 64, 96
    p.gray_shift = 14
    p.canny_low, p.canny_high = 20, 60
    p.lbp_points, p.lbp_radius = 16, 2.0
    op = coracle.default_params()
    op.fft_h, op.fft_w, op.gray_shift, op.canny_low, op.canny_high = 64, 96, 14, 20, 60
    op.lbp_points, op.lbp_radius = 16, 2.0
    r = ctx.analyze_frames([f], p)[0]
    _check_record(r, coracle.analyze_frame(f, op), P=16)


def test_gray_inside_stencil_pass_and_separate_gray_pass(ctx, coracle, monkeypatch):
    """Widths that are a multiple of four take the stencil pass that converts BGR itself (no separate gray pass);
    SIFT_TEXTURE_SEPARATE_GRAY=1 at context creation keeps the two passes apart.  Same results, both the oracle's,
    at both fixed-point precisions of the conversion, around every tile / strip / border case of the tile loader."""
    from sift_amd.texture import TextureContext
    monkeypatch.setenv("SIFT_TEXTURE_SEPARATE_GRAY", "1")
    sep = TextureContext(0)
    monkeypatch.delenv("SIFT_TEXTURE_SEPARATE_GRAY")
    try:
        shapes = [(3, 4), (4, 4), (3, 8), (5, 12), (64, 64), (127, 124), (129, 132), (130, 256), (131, 260), (257, 128),
                  (300, 392), (1080, 1920)]
        for i, (h, w) in enumerate(shapes):
            kind = KINDS[i % len(KINDS)]
            f = synth_frame(kind, SEED, 70 + i, 0, h, w)
            for shift in (14, 15):
                p = ctx.default_params()
                p.gray_shift = shift
                op = coracle.default_params()
                op.gray_shift = shift
                m = coracle.analyze_frame(f, op)
                a = ctx.analyze_frames([f], p)[0]
                b = sep.analyze_frames([f], p)[0]
                _check_record(a, m)
                _check_record(b, m)
                for k in O.METRIC_KEYS:
                    assert getattr(a, k) == getattr(b, k), (k, h, w, shift)
            # the Laplacian of the optional quality pass reads the plane's mirrored frame
            p = ctx.default_params()
            p.quality_stats = 1
            qa, qb = ctx.analyze_frames([f], p)[0], sep.analyze_frames([f], p)[0]
            want = coracle.quality_sums(coracle.gray(f))
            assert (qa.lap_sum, qa.lap_sumsq, qa.gray_sumsq) == want, (h, w)
            assert (qb.lap_sum, qb.lap_sumsq, qb.gray_sumsq) == want, (h, w)
            if h * w <= 300 * 392:
                ga, ca, ea = ctx.debug_planes(f)
                gb, cb, eb = sep.debug_planes(f)
                assert np.array_equal(ga, gb) and np.array_equal(ca, cb) and np.array_equal(ea, eb), (h, w)
        # a batch whose frames differ, in one call
        fr = [synth_frame(KINDS[j % len(KINDS)], SEED, 90 + j, j, 200, 264) for j in range(5)]
        for a, b, f in zip(ctx.analyze_frames(fr), sep.analyze_frames(fr), fr):
            _check_record(a, coracle.analyze_frame(f))
            _check_record(b, coracle.analyze_frame(f))
    finally:
        sep.close()


def test_per_kernel_timing_by_family(ctx):
    """stx_set_timing: everything, a chosen set of kernel families, nothing."""
    f = [synth_frame("natural", SEED, 50, i, 96, 128) for i in range(3)]
    ctx.get_timing(reset=True)
    ctx.set_timing(True)
    ctx.analyze_frames(f)
    ctx.set_timing(False)
    t = ctx.get_timing(reset=True)["kernels"]
    assert t["stencil"]["launches"] == 1 and t["fft_rows"]["launches"] == 1 and t["fft_cols"]["launches"] == 1
    assert t["hysteresis"]["launches"] >= 1 and t["stencil"]["ms"] > 0
    assert t["gray"]["launches"] == 0  # a width that is a multiple of four: the stencil pass converts the frames
    ctx.set_timing(True, kinds=("stencil", "fft_cols"))
    ctx.analyze_frames(f)
    ctx.set_timing(False)
    t = ctx.get_timing(reset=True)
    assert t["frames"] == 3
    assert {k for k, v in t["kernels"].items() if v["launches"]} == {"stencil", "fft_cols"}
    ctx.analyze_frames(f)
    assert all(v["launches"] == 0 for v in ctx.get_timing()["kernels"].values())


def test_odd_fft_sizes(ctx, coracle):
    # prime and mixed sizes exercise the generic butterfly: 17x23, 7*11=77, 2*3*5*7=210, 13
    for (h, w) in [(17, 23), (77, 13), (210, 66), (49, 121), (26, 34)]:
        f = synth_frame("noise", SEED, 31, 0, h, w)
        _check_record(ctx.analyze_frames([f])[0], coracle.analyze_frame(f))


@pytest.mark.parametrize("h,w", [(521, 66), (66, 523), (521, 523), (2 * 263, 3 * 257), (4099, 64), (48, 4099), (1, 263), (263, 1)])
def test_sizes_with_large_prime_factors(ctx, coracle, h, w):
    # lengths with a prime factor above 256 take the chirp-z transform (m = 1024 .. 16384), in fp32 and in fp64
    f = synth_frame("noise", SEED, 33, 0, h, w)
    ref = _np_freq(coracle.gray(f))
    assert ctx.analyze_frames([f])[0].frequency_anomaly == pytest.approx(ref, rel=REL, abs=1e-9)
    p = ctx.default_params(); p.fft_exact = 1
    assert ctx.analyze_frames([f], p)[0].frequency_anomaly == pytest.approx(ref, rel=1e-9, abs=1e-12)
    if h * w < 100000:
        g = synth_frame("ramp", SEED, 33, 0, h, w)  # near-zero bins: the fp64 re-transform must kick in by itself
        assert ctx.analyze_frames([g])[0].frequency_anomaly == pytest.approx(_np_freq(coracle.gray(g)), rel=REL, abs=1e-9)


def test_long_contours_hysteresis(ctx, coracle):
    # weak contours hundreds of pixels long hanging off a single strong seed: many
    # tile-to-tile propagation passes
    h, w = 200, 330
    g = np.full((h, w), 40, np.uint8)
    for y in range(4, h - 4, 6):  # serpentine of faint lines (gradient 4*12=48 < 50 alone...)
        g[y, 4:w - 4] = 58
    g[4:h - 4, 4] = 58
    g[4:h - 4, w - 5] = 58
    g[10:14, 10:14] = 255  # one strong blob touching the serpentine
    bgr = np.ascontiguousarray(np.repeat(g[..., None], 3, axis=2))
    gray, _, edges = ctx.debug_planes(bgr)
    assert np.array_equal(gray, g)
    want, nms, cnt = coracle.canny(g)
    assert np.array_equal(edges, want)
    assert ctx.analyze_frames([bgr])[0].edge_count == cnt


def test_one_long_snake_needs_thousands_of_rounds(ctx, coracle):
    # a single weak path that crosses the frame back and forth, hanging off one strong seed: the edge grows by
    # one tile per pass, far beyond the grid passes queued with a chunk and beyond one continuation launch;
    # the other frames of the batch finish at once
    h, w = 520, 700
    g = np.full((h, w), 40, np.uint8)
    ys = list(range(8, h - 8, 10))
    for k, y in enumerate(ys):
        g[y:y + 3, 8:w - 8] = 60  # a faint band: both of its sides are weak edges
        if k + 1 < len(ys):  # connect to the next band at alternating ends
            x = w - 11 if k % 2 == 0 else 8
            g[y:ys[k + 1] + 3, x:x + 3] = 60
    g[6:12, 8:14] = 255  # the seed, on the first band
    snake = np.ascontiguousarray(np.repeat(g[..., None], 3, axis=2))
    frames = [snake, synth_frame("noise", SEED, 5, 0, h, w), synth_frame("flat", SEED, 5, 1, h, w)]
    res = ctx.analyze_frames(frames)
    for f, r in zip(frames, res):
        assert r.edge_count == coracle.canny(coracle.gray(f))[2]
    _, _, edges = ctx.debug_planes(snake)
    assert np.array_equal(edges, coracle.canny(g)[0])
    # the path really is connected end to end: its last line is an edge
    assert edges[ys[-1] - 2:ys[-1] + 5, w // 2].any()


def test_constant_frames_spectrum_is_exact(det):
    # the FFT runs on gray minus the frame's own first gray level: a constant frame transforms to exact zeros,
    # as in the reference's fp64, and only bin (0,0) carries ln(v*H*W + 1)
    for (h, w) in ((48, 64), (120, 160), (97, 131)):
        for v in (0, 1, 77, 128, 255):
            f = np.full((h, w, 3), v, np.uint8)
            want = float(np.std(np.log(np.abs(np.fft.fft2(f[:, :, 0].astype(np.float64))) + 1)))
            got = det.analyze_frame(f)["frequency_anomaly"]
            assert got == pytest.approx(want, rel=1e-6, abs=1e-12), (h, w, v)


def _np_freq(gray):
    # passive_checker.py:120-123 as numpy computes it (complex128)
    return float(np.std(np.log(np.abs(np.fft.fft2(gray.astype(np.float64))) + 1)))


def test_structured_frames_are_transformed_in_fp64(ctx, coracle):
    """Frames whose exact spectrum is zero in many bins (ramps, bars, one edge on a constant) would show the fp32
    transform's rounding noise in ln(|F|+1); the library notices and repeats their transform in fp64, so the
    statistic holds the same tolerance as on camera-like content."""
    from sift_amd._lib import FLAG_FFT_F64
    h, w = 200, 328
    y, x = np.mgrid[0:h, 0:w]
    planes = {
        "ramp_x": (x & 255), "ramp_xy": ((x + 2 * y) & 255), "bars": 40 + 150 * ((x >> 5) & 1),
        "one_edge": np.where(x < 150, 90, 170), "disc": np.where((x - 160) ** 2 + (y - 100) ** 2 < 60 ** 2, 200, 50),
        "black": np.zeros((h, w), np.int64),
    }
    for name, g in planes.items():
        g = g.astype(np.uint8)
        f = np.ascontiguousarray(np.repeat(g[..., None], 3, axis=2))
        r = ctx.analyze_frames([f])[0]
        want = _np_freq(g)
        assert r.frequency_anomaly == pytest.approx(want, rel=1e-6, abs=1e-9), (name, bool(r.flags & FLAG_FFT_F64))
    # camera-like content stays on the fp32 path and still agrees
    for kind in ("noise", "natural"):
        f = synth_frame(kind, SEED, 50, 0, h, w)
        r = ctx.analyze_frames([f])[0]
        assert not (r.flags & FLAG_FFT_F64), kind
        assert r.frequency_anomaly == pytest.approx(_np_freq(coracle.gray(f)), rel=REL)
    # quiet frames: the noise of the fp32 transform scales with the input's energy, so a constant frame (no energy:
    # an exactly zero spectrum) and a two-level frame stay on the fp32 path and agree all the same
    for f in (np.full((h, w, 3), 93, np.uint8), synth_frame("tie", SEED, 50, 0, h, w), synth_frame("flat", SEED, 50, 0, h, w)):
        r = ctx.analyze_frames([f])[0]
        assert not (r.flags & FLAG_FFT_F64)
        assert r.frequency_anomaly == pytest.approx(_np_freq(coracle.gray(f)), rel=REL, abs=1e-9)
    # the knob: every frame in fp64 agrees with numpy to rounding; fp32 only never sets the flag
    f = synth_frame("natural", SEED, 50, 1, h, w)
    p = ctx.default_params(); p.fft_exact = 1
    r = ctx.analyze_frames([f], p)[0]
    assert r.flags & FLAG_FFT_F64
    assert r.frequency_anomaly == pytest.approx(_np_freq(coracle.gray(f)), rel=1e-10)
    p.fft_exact = -1
    f = np.ascontiguousarray(np.repeat(planes["ramp_x"].astype(np.uint8)[..., None], 3, axis=2))
    assert not (ctx.analyze_frames([f], p)[0].flags & FLAG_FFT_F64)


@pytest.mark.parametrize("h,w", [(17, 23), (77, 13), (210, 66), (1, 9), (7, 1), (96, 160), (250, 243)])
def test_fp64_transform_any_size(ctx, coracle, h, w):
    # generic radices, primes, single rows / columns through the fp64 kernels
    p = ctx.default_params(); p.fft_exact = 1
    for kind in ("noise", "ramp", "step"):
        f = synth_frame(kind, SEED, 51, 0, h, w)
        r = ctx.analyze_frames([f], p)[0]
        assert r.frequency_anomaly == pytest.approx(_np_freq(coracle.gray(f)), rel=1e-9, abs=1e-12), (kind, h, w)


def test_errors_are_python_exceptions(ctx, det):
    from sift_amd._lib import StxError
    with pytest.raises(ValueError):
        det.analyze_frame(np.zeros((8, 8), np.uint8))
    with pytest.raises(TypeError):
        det.analyze_frame(np.zeros((8, 8, 3), np.float32))
    p = ctx.default_params(); p.lbp_points = 40
    with pytest.raises(StxError):
        ctx.analyze_frames([np.zeros((8, 8, 3), np.uint8)], p)
    p = ctx.default_params(); p.fft_h = 100
    with pytest.raises(StxError):
        ctx.analyze_frames([np.zeros((8, 8, 3), np.uint8)], p)
    assert det.compute_texture_score([]) == {
        "passed": False, "average_score": 0.0, "individual_scores": [], "detailed_metrics": [],
        "message": "Texture analysis failed (score: 0.000)"}


def test_two_detectors_one_process():
    from sift_amd.texture import TextureNoiseDetector
    a, b = TextureNoiseDetector(), TextureNoiseDetector()
    f = synth_frame("natural", SEED, 40, 0, 64, 64)
    assert a.analyze_frame(f) == b.analyze_frame(f)
    assert a.threshold == 0.6


def test_async_submit_wait(ctx, coracle):
    import ctypes as C
    from sift_amd._lib import StxError, StxMetrics
    h, w, n = 96, 160, 6
    nbytes = n * h * w * 3
    devs = [ctx.device_alloc(nbytes) for _ in range(2)]
    try:
        for v, d in enumerate(devs):
            ctx.synth_device(d, KIND_ID["natural"], SEED + 7, v, 0, n, h, w)
        want = [ctx.analyze_device(d, n, h, w) for d in devs]
        assert ctx.max_async_frames(h, w) >= n
        outs = [(StxMetrics * n)() for _ in range(2)]
        for round_ in range(3):
            t0 = ctx.submit_device(devs[0], n, h, w, outs[0])
            t1 = ctx.submit_device(devs[1], n, h, w, outs[1])
            assert t0 != t1
            with pytest.raises(StxError):
                ctx.submit_device(devs[0], n, h, w, outs[0])  # only two in flight
            with pytest.raises(StxError):
                ctx.analyze_device(devs[0], n, h, w)          # blocking calls refuse to interleave
            ctx.wait(t1)
            ctx.wait(t0)
            with pytest.raises(StxError):
                ctx.wait(t0)
            for o, wv in zip(outs, want):
                assert bytes(o) == bytes(wv)[: C.sizeof(o)]
        m = coracle.analyze_frame(synth_clip("natural", SEED + 7, 1, n, h, w)[2])
        _check_record(outs[1][2], m)
    finally:
        for d in devs:
            ctx.device_free(d)
