"""The entry point that allocates nothing (stx_analyze_batch_device_noalloc) against stx_analyze_batch_device:
two calls in a row through the C ABI, the second with a smaller input (the static scratch array keeps the first
call's contents)."""
import ctypes as C

import numpy as np
import pytest

from sift_amd.synth import KIND_ID

pytestmark = pytest.mark.gpu

INT_FIELDS = ("edge_count", "absdiff_sum", "hp_sum", "hp_sumsq", "gray_sum", "n_pixels")
FLOAT_FIELDS = ("lbp_variance", "noise_level", "compression_artifacts", "edge_density", "frequency_anomaly",
                "texture_score")


def _noalloc(ctx, dev, n, h, w, params=None):
    from sift_amd._lib import StxMetrics
    out = (StxMetrics * n)()
    rc = ctx.lib.stx_analyze_batch_device_noalloc(ctx.handle, dev, n, h, w, 3 * w, 3 * w * h,
                                                  C.byref(params) if params is not None else None, out)
    ctx._check(rc, "stx_analyze_batch_device_noalloc")
    return out


def _same(a, b, n):
    for i in range(n):
        assert list(a[i].lbp_hist[:10]) == list(b[i].lbp_hist[:10]), i
        for f in INT_FIELDS:
            assert getattr(a[i], f) == getattr(b[i], f), (i, f)
        for f in FLOAT_FIELDS:
            assert getattr(a[i], f) == pytest.approx(getattr(b[i], f), rel=1e-4, abs=1e-12), (i, f)
        assert a[i].flags == b[i].flags


def test_noalloc_matches_allocating_entry_point_twice():
    from sift_amd._lib import StxError
    from sift_amd.texture import TextureContext
    ctx = TextureContext(0)
    try:
        # first call: more frames than one pass over the scratch array takes at this size, camera-like and
        # structured content (the second goes through the fp64 transform); second call: fewer, smaller frames
        for (n, h, w, kind) in ((70, 270, 480, "natural"), (5, 1080, 1920, "natural"), (3, 200, 328, "ramp"),
                                (2, 96, 160, "noise"),
                                # lengths with a large prime factor: the chirp-z tables go through the scratch array too
                                (2, 263, 523, "noise"), (1, 263, 523, "ramp")):
            dev = ctx.device_alloc(n * h * w * 3)
            try:
                ctx.synth_device(dev, KIND_ID[kind], 0x5EED0009, 1, 0, n, h, w)
                got = _noalloc(ctx, dev, n, h, w)
                want = ctx.analyze_device(dev, n, h, w)
                _same(got, want, n)
                again = _noalloc(ctx, dev, n, h, w)
                assert bytes(again) == bytes(got)
            finally:
                ctx.device_free(dev)
        # beyond the maximum: a clear error, not an allocation
        h, w = 4096, 8192
        dev = ctx.device_alloc(h * w * 3)
        try:
            with pytest.raises(StxError, match="too large for the static scratch"):
                _noalloc(ctx, dev, 1, h, w)
        finally:
            ctx.device_free(dev)
        p = ctx.default_params(); p.lbp_points = 16; p.lbp_radius = 2.0
        dev = ctx.device_alloc(64 * 64 * 3)
        try:
            with pytest.raises(StxError):
                _noalloc(ctx, dev, 1, 64, 64, p)
        finally:
            ctx.device_free(dev)
    finally:
        ctx.close()
