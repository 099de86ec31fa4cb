"""Frame preparation of the CNN branch (passive_checker.py:393-418): the device kernel against the
oracle's restatement of cv2.resize's 8-bit fixed-point bilinear arithmetic.

Pins: no OpenCV in this image and no fixture of resized frames in the reference, so the resize is
"parity unpinned" as far as the reference is concerned.  The restatement is held to the properties
OpenCV documents or that follow from its arithmetic: identity at equal sizes, constants stay constant,
hand-computed 2x enlargement, the 2:1 area rule, and at most one gray level from an exact
half-pixel-centre bilinear blend.
"""
import numpy as np
import pytest

from oracle.texture_oracle import np_preprocess_frame, np_resize_linear_u8
from sift_amd.synth import synth_frame

SEED = 0x5EED0000


def _float_bilinear(img, out_h, out_w):
    h, w = img.shape[:2]
    a = img.astype(np.float64)
    ys = np.clip((np.arange(out_h) + 0.5) * h / out_h - 0.5, 0, h - 1)
    xs = np.clip((np.arange(out_w) + 0.5) * w / out_w - 0.5, 0, w - 1)
    y0 = np.floor(ys).astype(int); y1 = np.minimum(y0 + 1, h - 1); fy = (ys - y0)[:, None, None]
    x0 = np.floor(xs).astype(int); x1 = np.minimum(x0 + 1, w - 1); fx = (xs - x0)[None, :, None]
    a = a.reshape(h, w, -1)
    top = a[y0][:, x0] * (1 - fx) + a[y0][:, x1] * fx
    bot = a[y1][:, x0] * (1 - fx) + a[y1][:, x1] * fx
    return (top * (1 - fy) + bot * fy).reshape((out_h, out_w) + img.shape[2:])


def test_resize_known_answers():
    rng = np.random.default_rng(3)
    a = rng.integers(0, 256, (80, 80, 3), dtype=np.uint8)
    assert (np_resize_linear_u8(a, 80, 80) == a).all()
    assert (np_resize_linear_u8(np.full((480, 640, 3), 77, np.uint8), 80, 80) == 77).all()
    assert (np_resize_linear_u8(np.full((3, 5), 255, np.uint8), 80, 80) == 255).all()
    # 2x enlargement of one row: taps at -0.25, 0.25, 0.75, 1.25 -> border, 3:1, 1:3, border
    row = np.array([[200, 100]], np.uint8)
    assert np_resize_linear_u8(row, 1, 4).tolist() == [[200, 175, 125, 100]]
    # exact 2:1 reduction averages 2x2 blocks with rounding
    b = np.array([[1, 2, 10, 20], [3, 4, 30, 41]], np.uint8)
    assert np_resize_linear_u8(b, 1, 2).tolist() == [[3, 25]]


@pytest.mark.parametrize("shape,out", [((480, 640), (80, 80)), ((1080, 1920), (80, 80)), ((37, 53), (80, 80)),
                                       ((200, 100), (64, 96)), ((90, 90), (80, 80))])
def test_resize_close_to_exact_bilinear(shape, out):
    rng = np.random.default_rng(shape[0])
    img = rng.integers(0, 256, shape + (3,), dtype=np.uint8)
    got = np_resize_linear_u8(img, out[0], out[1]).astype(np.float64)
    ref = _float_bilinear(img, out[0], out[1])
    assert np.abs(got - ref).max() <= 1.0


def test_preprocess_layout():
    f = synth_frame("natural", SEED, 0, 0, 120, 160)
    t = np_preprocess_frame(f)
    assert t.shape == (1, 3, 80, 80) and t.dtype == np.float32
    assert 0.0 <= t.min() and t.max() <= 1.0
    # plane 0 is red = channel 2 of the BGR frame
    r = np_resize_linear_u8(f[:, :, 2], 80, 80).astype(np.float32) / 255.0
    assert (t[0, 0] == r).all()


GPU_CASES = [((1080, 1920), (80, 80)), ((480, 640), (80, 80)), ((720, 1280), (80, 80)), ((77, 131), (80, 80)),
             ((160, 160), (80, 80)), ((80, 80), (80, 80)), ((1, 1), (80, 80)), ((2, 3), (5, 7)),
             ((333, 517), (224, 224)), ((2160, 3840), (80, 80))]


@pytest.mark.gpu
@pytest.mark.parametrize("shape,out", GPU_CASES)
def test_gpu_preprocess_matches_oracle(shape, out):
    from sift_amd.texture import TextureContext
    ctx = TextureContext(0)
    h, w = shape
    frames = [synth_frame(k, SEED + 5, 0, i, h, w) for i, k in enumerate(("natural", "noise", "ramp"))]
    got = ctx.preprocess_frames(frames, out_h=out[0], out_w=out[1])
    assert got.shape == (3, 3, out[0], out[1]) and got.dtype == np.float32
    for i, f in enumerate(frames):
        want = np_preprocess_frame(f, (out[1], out[0]))[0]
        assert np.array_equal(got[i], want), (shape, out, i)


@pytest.mark.gpu
def test_gpu_preprocess_device_strided_and_class():
    import torch
    from sift_amd.preprocess import FramePreprocessor
    from sift_amd.texture import TextureContext
    h, w, n = 96, 150, 4
    frames = [synth_frame("natural", SEED + 6, 1, i, h, w) for i in range(n)]
    # device-resident frames with padded rows and padded frames
    rs, fs = 3 * w + 10, (3 * w + 10) * h + 64
    buf = np.zeros(n * fs, np.uint8)
    for i, f in enumerate(frames):
        v = buf[i * fs:i * fs + rs * h].reshape(h, rs)
        v[:, :3 * w] = f.reshape(h, 3 * w)
    dev = torch.from_numpy(buf).cuda()
    out = torch.empty((n, 3, 80, 80), dtype=torch.float32, device="cuda")
    ctx = TextureContext(0)
    ctx.preprocess_device(dev.data_ptr(), n, h, w, out.data_ptr(), row_stride=rs, frame_stride=fs)
    want = np.concatenate([np_preprocess_frame(f) for f in frames])
    assert np.array_equal(out.cpu().numpy(), want)
    # the class with the reference's method name
    pre = FramePreprocessor()
    assert pre.input_size == (80, 80)
    t = pre._preprocess_frame(frames[0])
    assert t.shape == (1, 3, 80, 80) and t.dtype == torch.float32 and t.is_cuda
    assert np.array_equal(t.cpu().numpy(), want[:1])
    with pytest.raises(ValueError):
        pre._preprocess_frame(np.zeros((4, 4), np.uint8))
    # argument checks come back as errors, not crashes
    from sift_amd._lib import StxError
    with pytest.raises(StxError):
        ctx.preprocess_device(dev.data_ptr(), n, h, w, out.data_ptr(), out_h=0)
    with pytest.raises(StxError):
        ctx.preprocess_device(dev.data_ptr(), n, h, w, out.data_ptr(), row_stride=3 * w - 1)
