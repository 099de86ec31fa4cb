"""Every compile-time-planned FFT length, as a row transform and as a column transform, against
numpy (np.fft.fft2 -> log(abs+1) -> std, /root/reference/app/passive_checker.py:120-123)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

FAST_SIZES = [64, 80, 96, 128, 160, 240, 256, 320, 360, 480, 512, 540, 640, 720, 768, 960, 1024, 1080, 1280, 1440,
              1920, 2048, 2160, 2560, 3840, 4096]


@pytest.fixture(scope="module")
def ctx():
    from sift_amd.texture import TextureContext
    c = TextureContext(0)
    yield c
    c.close()


def _ref(gray):
    return float(np.std(np.log(np.abs(np.fft.fft2(gray)) + 1)))


@pytest.mark.parametrize("n", FAST_SIZES)
def test_fast_fft_length(ctx, n):
    rng = np.random.default_rng(n)
    # 66 has no planned kernel, so each case isolates one planned transform; the square case runs both
    for (h, w) in ((n, 66), (66, n), (n, n) if n <= 512 else (n, 64)):
        g = rng.integers(0, 256, (h, w), dtype=np.uint8)
        bgr = np.ascontiguousarray(np.repeat(g[..., None], 3, axis=2))
        r = ctx.analyze_frames([bgr, bgr[::-1].copy()])
        assert r[0].frequency_anomaly == pytest.approx(_ref(g), rel=1e-5), (h, w)
        assert r[1].frequency_anomaly == pytest.approx(_ref(g[::-1]), rel=1e-5), (h, w)


def test_4k_full_frame_transform(ctx):
    """The default call on a 4K frame transforms the whole frame (passive_checker.py:120): 2160 x 3840."""
    from sift_amd.synth import synth_frame
    f = synth_frame("natural", 0x5EED0004, 1, 0, 2160, 3840)
    r = ctx.analyze_frames([f])[0]
    # the gray plane as the library computes it, through the debug entry point, then numpy's complex128 transform
    gray, _, _ = ctx.debug_planes(f)
    assert r.frequency_anomaly == pytest.approx(_ref(gray), rel=1e-5)


@pytest.mark.parametrize("hw", [(67, 1920), (67, 1280), (33, 640), (135, 960), (1081, 1920)])
def test_row_pairs_with_an_odd_height(ctx, hw):
    """The row pass transforms two image rows at a time and writes both rows' bins of a column block as one line;
    the last row of an odd-height frame has no partner, and the next frame's first line follows it directly."""
    h, w = hw
    rng = np.random.default_rng(h * 4096 + w)
    frames = []
    for _ in range(3):
        g = rng.integers(0, 256, (h, w), dtype=np.uint8)
        frames.append((g, np.ascontiguousarray(np.repeat(g[..., None], 3, axis=2))))
    r = ctx.analyze_frames([b for _, b in frames])
    for (g, _), m in zip(frames, r):
        assert m.frequency_anomaly == pytest.approx(_ref(g), rel=1e-5), hw


@pytest.mark.parametrize("hw", [(540, 256), (540, 128), (540, 96), (360, 160), (540, 64)])
def test_sub_wave_rows_on_sixteen_bin_blocks(ctx, hw):
    """Row transforms that share a wave (widths up to 256) write all their rows of a sixteen-bin column block with one
    instruction; 540 rows leave the last wave of a frame with half of its row pairs."""
    h, w = hw
    rng = np.random.default_rng(h * 4096 + w)
    frames = []
    for _ in range(3):
        g = rng.integers(0, 256, (h, w), dtype=np.uint8)
        frames.append((g, np.ascontiguousarray(np.repeat(g[..., None], 3, axis=2))))
    r = ctx.analyze_frames([b for _, b in frames])
    for (g, _), m in zip(frames, r):
        assert m.frequency_anomaly == pytest.approx(_ref(g), rel=1e-5), hw
