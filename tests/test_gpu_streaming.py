"""The pieces of a job whose clips exist only while they are in flight: the device generator's fast kernel against
the numpy generator, its asynchronous form ordered before the analysis that follows, the asynchronous extended entry
point against the blocking one, and bench.py --config clips10k at a small frame size (video ids beyond the clips a rank
could keep resident, spot-checked against regenerated frames and the oracle)."""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from sift_amd.synth import KIND_ID, synth_clip

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SEED = 0x5EED0000


@pytest.fixture(scope="module")
def ctx():
    from sift_amd.texture import TextureContext
    c = TextureContext(0)
    yield c
    c.close()


@pytest.mark.parametrize("h,w,n,video,frame0", [
    (96, 160, 3, 5, 3),        # one partly filled wave per row
    (67, 256, 2, 9, 0),        # exactly one wave, a height that is not a multiple of the rows per wave
    (19, 260, 2, 1 << 20, 7),  # one full wave and one lane of the next
    (41, 1284, 1, 77, 1000),   # several waves; late frame index (moving centre far from the middle)
    (8, 4, 2, 3, 0),           # one lane
    (33, 122, 2, 4, 5),        # width not a multiple of four: the per-pixel kernel
    (270, 480, 4, 51, 146),
])
def test_device_generator_equals_numpy_generator(ctx, h, w, n, video, frame0):
    nbytes = n * h * w * 3
    dev = ctx.device_alloc(nbytes)
    try:
        ctx.synth_device(dev, KIND_ID["natural"], SEED + 3, video, frame0, n, h, w)
        got = ctx.d2h(dev, nbytes).reshape(n, h, w, 3)
        want = synth_clip("natural", SEED + 3, video, n, h, w, frame0=frame0)
        assert np.array_equal(got, want)
    finally:
        ctx.device_free(dev)


def test_async_generator_is_ordered_before_the_next_analysis(ctx):
    from sift_amd._lib import StxMetrics
    h, w, n = 270, 480, 12
    nbytes = n * h * w * 3
    bufs = [ctx.device_alloc(nbytes) for _ in range(3)]
    outs = [(StxMetrics * n)() for _ in range(2)]
    try:
        want = {}
        for v in range(6):
            ctx.synth_device(bufs[0], KIND_ID["natural"], SEED + 3, v, 0, n, h, w)
            want[v] = bytes(ctx.analyze_device(bufs[0], n, h, w))
        # the streaming loop: generate into a rotating buffer, submit, collect the clip before
        got, pending = {}, None
        for v in range(6):
            ctx.synth_device_async(bufs[v % 3], KIND_ID["natural"], SEED + 3, v, 0, n, h, w)
            t = ctx.submit_device(bufs[v % 3], n, h, w, outs[v % 2])
            if pending is not None:
                ctx.wait(pending[0])
                got[pending[1]] = bytes(outs[pending[1] % 2])
            pending = (t, v)
        ctx.wait(pending[0])
        got[pending[1]] = bytes(outs[pending[1] % 2])
        assert got == want
        # and before a blocking call
        ctx.synth_device_async(bufs[1], KIND_ID["natural"], SEED + 3, 2, 0, n, h, w)
        assert bytes(ctx.analyze_device(bufs[1], n, h, w)) == want[2]
    finally:
        for b in bufs:
            ctx.device_free(b)


def test_async_extended_equals_blocking_extended(ctx):
    from sift_amd._lib import StxExtended, StxMetrics
    h, w, n = 200, 328, 5
    nbytes = n * h * w * 3
    devs = [ctx.device_alloc(nbytes) for _ in range(2)]
    try:
        want = []
        for k, d in enumerate(devs):
            ctx.synth_device(d, KIND_ID["natural"], SEED + 4, k, 0, n, h, w)
            m, x = ctx.analyze_device_extended(d, n, h, w)
            want.append((bytes(m), bytes(x)))
        outs = [(StxMetrics * n)() for _ in devs]
        exts = [(StxExtended * n)() for _ in devs]
        tickets = [ctx.submit_device_extended(d, n, h, w, outs[k], exts[k]) for k, d in enumerate(devs)]
        for t in tickets:
            ctx.wait(t)
        for k in range(2):
            assert (bytes(outs[k]), bytes(exts[k])) == want[k]
        # a plain submission into a slot that last held an extended one computes no extended statistics
        t = ctx.submit_device(devs[0], n, h, w, outs[0])
        ctx.wait(t)
        assert bytes(outs[0]) == want[0][0]
    finally:
        for d in devs:
            ctx.device_free(d)


def test_streaming_bench_small_frames():
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "clips10k", "--steps", "60",
                          "--warmup", "1", "--frames", "6", "--height", "270", "--width", "480"],
                         env=env, timeout=300, capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-2000:]
    r = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][0])
    st = r["streaming"]
    assert st["clips_total"] == 60 and r["config"]["clips_resident_per_gpu"] == 3
    sc = st["spot_check"]
    assert sc["ok"] and sc["max_rel_diff_vs_oracle"] <= 1e-4
    assert min(v for v, _ in sc["video_frame_pairs"]) >= 51
    assert st["synth_kernel_ms_per_clip_in_situ"] > 0
