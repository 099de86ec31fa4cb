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
    """Both entry points run the same kernels on the same frames and sum in a fixed order: every field is equal,
    floats included."""
    for i in range(n):
        assert list(a[i].lbp_hist[:10]) == list(b[i].lbp_hist[:10]), i
        for f in INT_FIELDS:
            assert getattr(a[i], f) == getattr(b[i], f), (i, f)
        for f in FLOAT_FIELDS:
            assert getattr(a[i], f) == getattr(b[i], f), (i, f)
        assert a[i].flags == b[i].flags
    assert bytes(a) == bytes(b)


def _free_device_bytes():
    """Free device memory as the HIP runtime the library is linked against reports it."""
    hip = C.CDLL("libamdhip64.so")
    free, total = C.c_size_t(0), C.c_size_t(0)
    assert hip.hipMemGetInfo(C.byref(free), C.byref(total)) == 0
    return free.value


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
                before = _free_device_bytes()
                again = _noalloc(ctx, dev, n, h, w)
                assert _free_device_bytes() == before, "the allocation-free entry point changed free device memory"
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


def test_noalloc_contexts_of_one_gpu_share_the_tables():
    """The twiddle tables live in the GPU's scratch array, which every context of that GPU uses: context A at one
    size, context B at another, A again at the first size must not transform with B's tables."""
    from sift_amd.texture import TextureContext
    a, b = TextureContext(0), TextureContext(0)
    try:
        big, small = (3, 270, 480), (2, 96, 160)
        devs = {}
        for ctx, (n, h, w) in ((a, big), (b, small)):
            devs[ctx] = ctx.device_alloc(n * h * w * 3)
            ctx.synth_device(devs[ctx], KIND_ID["natural"], 0x5EED000A, 2, 0, n, h, w)
        want_a = a.analyze_device(devs[a], *big)
        want_b = b.analyze_device(devs[b], *small)
        _same(_noalloc(a, devs[a], *big), want_a, big[0])
        _same(_noalloc(b, devs[b], *small), want_b, small[0])
        _same(_noalloc(a, devs[a], *big), want_a, big[0])
        _same(_noalloc(b, devs[b], *small), want_b, small[0])
    finally:
        for ctx, d in devs.items():
            ctx.device_free(d)
        a.close()
        b.close()


def test_noalloc_overlapping_calls_on_one_gpu_are_refused_not_raced():
    """Two threads, a context each, calling at the same time: a call either gets the GPU's scratch array to itself
    and returns the right records, or is refused with a clear error."""
    import threading
    import time

    from sift_amd._lib import StxError
    from sift_amd.texture import TextureContext
    ctxs = [TextureContext(0), TextureContext(0)]
    shapes = [(24, 270, 480), (16, 200, 328)]
    devs, wants = [], []
    try:
        for ctx, (n, h, w) in zip(ctxs, shapes):
            d = ctx.device_alloc(n * h * w * 3)
            devs.append(d)
            ctx.synth_device(d, KIND_ID["natural"], 0x5EED000B, 3, 0, n, h, w)
            wants.append(bytes(ctx.analyze_device(d, n, h, w)))
        done, refused, wrong, errors = [0, 0], [0, 0], [0, 0], []

        def work(k):
            ctx, (n, h, w) = ctxs[k], shapes[k]
            deadline = time.monotonic() + 60.0
            while done[k] < 30 and time.monotonic() < deadline:
                try:
                    got = _noalloc(ctx, devs[k], n, h, w)
                except StxError as e:
                    if "another allocation-free call" in str(e):
                        refused[k] += 1
                        time.sleep(0.0002)
                    else:
                        errors.append(str(e))
                    continue
                done[k] += 1
                if bytes(got) != wants[k]:
                    wrong[k] += 1

        threads = [threading.Thread(target=work, args=(k,)) for k in range(2)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        assert not errors, errors
        assert wrong == [0, 0], (wrong, done, refused)
        assert done == [30, 30], (done, refused)
    finally:
        for ctx, d in zip(ctxs, devs):
            ctx.device_free(d)
        for ctx in ctxs:
            ctx.close()
