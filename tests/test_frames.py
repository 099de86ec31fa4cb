"""Frame-source contract (sift_amd/frames.py): the extract_frames layout in pinned memory."""
import io

import numpy as np
import pytest

from sift_amd.synth import synth_clip

SEED = 0x5EED0000


def test_frame_lists_with_mixed_row_strides():
    """A crop view (padded rows) beside a contiguous frame: the reference takes any mix (each frame goes through
    cv2.cvtColor on its own, passive_checker.py:100); here one call has one row stride, so a mix is made contiguous."""
    from sift_amd.texture import TextureContext
    big = synth_clip("noise", SEED, 1, 1, 40, 60)[0]
    view = big[4:24, 10:40]                      # strides (180, 3, 1)
    cont = np.ascontiguousarray(big[0:20, 0:30])  # strides (90, 3, 1)
    for frames in ([view, cont], [cont, view], [view, view], [cont, cont[:, ::-1]]):
        keep, ptrs, h, w, stride = TextureContext._frame_ptrs(frames)
        assert (h, w) == (20, 30) and len(keep) == 2
        assert all(f.strides == (stride, 3, 1) for f in keep)
        assert all(np.array_equal(k, f) for k, f in zip(keep, frames))
    keep, _, _, _, stride = TextureContext._frame_ptrs([view, view])
    assert stride == 180 and keep[0] is view  # equal padded strides pass without a copy


@pytest.mark.gpu
def test_views_outlive_the_batch():
    import gc
    from sift_amd.frames import PinnedFrameBatch
    from sift_amd.texture import TextureContext
    ctx = TextureContext(0)
    batch = PinnedFrameBatch(ctx, 3, 16, 24)
    clip = synth_clip("noise", SEED + 6, 0, 3, 16, 24)
    for f in clip:
        batch.append(f)
    frames = batch.frames()
    batch.close()
    del batch
    gc.collect()
    # the page-locked memory is still there: the views own it
    assert np.array_equal(np.stack(frames), clip)
    del frames
    gc.collect()
    ctx.close()


@pytest.mark.gpu
def test_pinned_batch_roundtrip_and_analysis():
    from sift_amd.frames import PinnedFrameBatch
    from sift_amd.texture import TextureContext, TextureNoiseDetector
    ctx = TextureContext(0)
    h, w, n = 45, 77, 7
    clip = synth_clip("natural", SEED + 5, 2, n, h, w)
    raw = io.BytesIO(np.ascontiguousarray(clip).tobytes() + b"\x01\x02\x03")  # plus a truncated frame
    batch = PinnedFrameBatch(ctx, 10, h, w)
    assert batch.read_raw_bgr(raw) == n
    frames = batch.frames()
    assert len(frames) == n and all(f.shape == (h, w, 3) and f.flags["C_CONTIGUOUS"] for f in frames)
    assert np.array_equal(np.stack(frames), clip)
    # the pinned batch scores exactly like a list of ordinary arrays
    recs = batch.analyze()
    want = ctx.analyze_frames([clip[i] for i in range(n)])
    for a, b in zip(recs, want):
        assert a.texture_score == b.texture_score and a.edge_count == b.edge_count and a.hp_sumsq == b.hp_sumsq
    det = TextureNoiseDetector()
    assert det.compute_texture_score(frames) == det.compute_texture_score([clip[i] for i in range(n)])
    # max_frames and capacity
    batch.clear()
    assert batch.read_raw_bgr(io.BytesIO(clip.tobytes()), max_frames=3) == 3
    batch.clear()
    small = PinnedFrameBatch(ctx, 2, h, w)
    assert small.read_raw_bgr(io.BytesIO(clip.tobytes())) == 2
    assert small.append(clip[0]) is False
    small.clear()
    assert small.append(clip[4]) is True and np.array_equal(small.frames()[0], clip[4])
    with pytest.raises(ValueError):
        small.append(np.zeros((h, w), np.uint8))
    small.close()
    batch.close()
    ctx.close()
