"""Extended statistics (multi-radius LBP histograms, Sobel statistics, 8x8 blockiness).

The reference computes none of these, so there is nothing of its own to pin them to: the definitions are in
include/sift_texture.h, the oracle restates them with numpy/scipy, hand-computed cases pin the oracle, and
the LBP histograms at (16, 2) and (24, 3) are the scikit-image ones the goldens already cover.
"""
import numpy as np
import pytest

from oracle.texture_oracle import np_canny, np_extended_stats, np_extended_sums, np_gray, np_lbp_uniform
from sift_amd.synth import synth_frame

SEED = 0x5EED0000


def test_extended_sums_known_answers():
    flat = np.full((16, 24), 9, np.uint8)
    s = np_extended_sums(flat)
    assert s["sobel_sum"] == 0 and s["sobel_strong"] == 0 and s["block_sums"] == [0, 0, 0, 0]
    # 16 rows; x = 1..23 holds boundaries at 8 and 16
    assert s["block_counts"] == [16 * 2, 16 * 21, 24 * 1, 24 * 14]
    # vertical stripes of width 8: every horizontal boundary difference is 10, nothing inside blocks
    img = np.zeros((8, 24), np.uint8)
    img[:, 8:16] = 10
    s = np_extended_sums(img)
    assert s["block_sums"] == [8 * 2 * 10, 0, 0, 0]
    # Sobel of a vertical step of height 10: |gx| = 40 on the two columns next to each step, gy = 0
    assert s["sobel_sum"] == 8 * 4 * 40 and s["sobel_sumsq"] == 8 * 4 * 1600
    assert s["sobel_strong"] == 0 and np_extended_sums(img, high=39)["sobel_strong"] == 8 * 4
    st = np_extended_stats(img, edge_count=0)
    assert st["blockiness"] == pytest.approx((160 / (8 * 2 + 0)) / 1e-10)
    # the same stripes shifted by 4 pixels are all "interior"
    img = np.zeros((8, 24), np.uint8)
    img[:, 4:12] = 10
    assert np_extended_sums(img)["block_sums"] == [0, 8 * 2 * 10, 0, 0]


GPU_SIZES = [(64, 64), (97, 131), (1, 1), (1, 40), (33, 1), (8, 8), (9, 17), (240, 320), (300, 390),
             # around the 256 columns of a wave and the four pixels of a lane in the Sobel / blockiness pass
             (5, 255), (5, 256), (5, 257), (20, 258), (17, 259), (40, 513), (3, 2), (2, 3), (70, 5), (66, 6), (65, 7)]


@pytest.mark.gpu
@pytest.mark.parametrize("shape", GPU_SIZES)
def test_gpu_extended_matches_oracle(shape):
    from sift_amd.extended import ExtendedTextureAnalyzer
    from sift_amd.texture import TextureNoiseDetector
    h, w = shape
    frames = [synth_frame(k, SEED + 11, 0, i, h, w) for i, k in enumerate(("natural", "noise", "step", "flat"))]
    res = ExtendedTextureAnalyzer().analyze_frames(frames)
    det = TextureNoiseDetector()
    for f, r in zip(frames, res):
        g = np_gray(f)
        edges, _ = np_canny(g, 50, 150)
        want = np_extended_stats(g, int(np.count_nonzero(edges)))
        for k in ("sobel_sum", "sobel_sumsq", "sobel_strong", "block_sums", "block_counts"):
            assert r[k] == want[k], (shape, k)
        for k in ("sobel_mean", "sobel_std", "strong_gradient_fraction", "edge_consistency", "blockiness"):
            assert r[k] == pytest.approx(want[k], rel=1e-12, abs=1e-300), (shape, k)
        # the reference's dict is untouched by the extended call
        assert r["metrics"] == det.analyze_frame(f)
        for key, P, R in (("P8_R1", 8, 1.0), ("P16_R2", 16, 2.0), ("P24_R3", 24, 3.0)):
            codes = np_lbp_uniform(g, P, R)
            assert r["lbp_histograms"][key] == np.bincount(codes.ravel().astype(np.int64), minlength=P + 2).tolist(), (shape, key)


@pytest.mark.gpu
def test_gpu_extended_device_batch_1080p():
    from sift_amd.texture import TextureContext
    ctx = TextureContext(0)
    h, w, n = 1080, 1920, 3
    dev = ctx.device_alloc(n * h * w * 3)
    try:
        ctx.synth_device(dev, 1, SEED + 12, 0, 0, n, h, w)
        out, ext = ctx.analyze_device_extended(dev, n, h, w)
        bgr = ctx.d2h(dev, n * h * w * 3).reshape(n, h, w, 3)
        for i in range(n):
            g = np_gray(bgr[i])
            want = np_extended_sums(g)
            assert int(ext[i].sobel_sum) == want["sobel_sum"] and int(ext[i].sobel_sumsq) == want["sobel_sumsq"]
            assert int(ext[i].sobel_strong) == want["sobel_strong"]
            assert [int(v) for v in ext[i].block_sums] == want["block_sums"]
            assert [int(v) for v in ext[i].block_counts] == want["block_counts"]
            assert sum(ext[i].lbp_hist_p16r2) == h * w and sum(ext[i].lbp_hist_p24r3) == h * w
    finally:
        ctx.device_free(dev)
