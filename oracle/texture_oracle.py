"""CPU oracle for the texture/noise hot path — TEST INFRASTRUCTURE ONLY.

Two things live here:

* ``COracle`` — a ctypes binding of ``oracle/libtexture_oracle.so`` (the C
  restatement in ``texture_oracle.c``).
* ``np_*`` — a second, independent restatement in numpy/scipy that shares no
  code with the C one.  Integer stages of the two must agree bit for bit
  (``tests/test_oracle.py``); that double implementation is the only check
  available for the OpenCV stages, which are PARITY UNPINNED (no OpenCV offline,
  no fixtures in the reference).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s cpu_baseline leg
may import this module.  The product package ``sift_amd`` never does.

Follows /root/reference/app/passive_checker.py:96-164.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libtexture_oracle.so")

LBP_MAX_BINS = 26


class OrcParams(C.Structure):
    _fields_ = [
        ("lbp_points", C.c_int),
        ("lbp_radius", C.c_double),
        ("canny_low", C.c_int),
        ("canny_high", C.c_int),
        ("gray_shift", C.c_int),
        ("fft_h", C.c_int),
        ("fft_w", C.c_int),
    ]


class OrcMetrics(C.Structure):
    _fields_ = [
        ("lbp_variance", C.c_double),
        ("noise_level", C.c_double),
        ("compression_artifacts", C.c_double),
        ("edge_density", C.c_double),
        ("frequency_anomaly", C.c_double),
        ("texture_score", C.c_double),
        ("lbp_hist", C.c_uint32 * LBP_MAX_BINS),
        ("edge_count", C.c_uint32),
        ("reserved", C.c_uint32),
        ("absdiff_sum", C.c_uint64),
        ("hp_sum", C.c_uint64),
        ("hp_sumsq", C.c_uint64),
        ("gray_sum", C.c_uint64),
        ("n_pixels", C.c_uint64),
        ("fft_sum", C.c_double),
        ("fft_sumsq", C.c_double),
    ]


def build(force: bool = False) -> str:
    """Compile the C oracle with gcc if the shared object is missing or stale."""
    src = os.path.join(_HERE, "texture_oracle.c")
    hdr = os.path.join(_HERE, "texture_oracle.h")
    stale = (
        force
        or not os.path.exists(_LIB)
        or (os.path.exists(src) and os.path.getmtime(_LIB) < max(os.path.getmtime(src), os.path.getmtime(hdr)))
    )
    if stale:
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B", "libtexture_oracle.so"])
    return _LIB


_u8p = C.POINTER(C.c_uint8)


def _p(a: np.ndarray):
    return a.ctypes.data_as(_u8p)


class COracle:
    """ctypes view of the C restatement."""

    def __init__(self):
        self.lib = C.CDLL(build())
        L = self.lib
        L.orc_default_params.argtypes = [C.POINTER(OrcParams)]
        L.orc_bgr2gray.argtypes = [_u8p, C.c_int, C.c_int, C.c_int64, C.c_int, _u8p]
        L.orc_lbp_offsets.argtypes = [C.c_int, C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.orc_lbp_uniform.argtypes = [_u8p, C.c_int, C.c_int, C.c_int, C.c_double, _u8p, C.POINTER(C.c_uint32)]
        L.orc_highpass3.argtypes = [_u8p, C.c_int, C.c_int, _u8p]
        L.orc_gauss5.argtypes = [_u8p, C.c_int, C.c_int, _u8p]
        L.orc_quality_sums.argtypes = [_u8p, C.c_int, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_uint64),
                                       C.POINTER(C.c_uint64)]
        L.orc_canny.argtypes = [_u8p, C.c_int, C.c_int, C.c_int, C.c_int, _u8p, _u8p]
        L.orc_fft_logmag_stats.argtypes = [
            _u8p, C.c_int, C.c_int, C.c_int64, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.orc_analyze_frame.argtypes = [_u8p, C.c_int, C.c_int, C.c_int64, C.POINTER(OrcParams), C.POINTER(OrcMetrics)]
        L.orc_synth_frame.argtypes = [C.c_int, C.c_uint64, C.c_uint32, C.c_uint32, C.c_int, C.c_int, _u8p]
        L.orc_texture_score.argtypes = [C.POINTER(OrcMetrics)]
        L.orc_texture_score.restype = C.c_double

    def default_params(self) -> OrcParams:
        p = OrcParams()
        self.lib.orc_default_params(C.byref(p))
        return p

    def gray(self, bgr: np.ndarray, shift: int = 15) -> np.ndarray:
        bgr = np.ascontiguousarray(bgr, dtype=np.uint8)
        h, w, _ = bgr.shape
        out = np.empty((h, w), np.uint8)
        self.lib.orc_bgr2gray(_p(bgr), h, w, w * 3, shift, _p(out))
        return out

    def lbp_offsets(self, P: int, R: float):
        rp = (C.c_double * 24)()
        cp = (C.c_double * 24)()
        if self.lib.orc_lbp_offsets(P, R, rp, cp) != 0:
            raise ValueError("bad P")
        return np.array(rp[:P]), np.array(cp[:P])

    def lbp(self, gray: np.ndarray, P: int = 8, R: float = 1.0):
        gray = np.ascontiguousarray(gray, dtype=np.uint8)
        h, w = gray.shape
        codes = np.empty((h, w), np.uint8)
        hist = (C.c_uint32 * LBP_MAX_BINS)()
        if self.lib.orc_lbp_uniform(_p(gray), h, w, P, R, _p(codes), hist) != 0:
            raise ValueError("bad P")
        return codes, np.array(hist[: P + 2], dtype=np.uint32)

    def highpass(self, gray: np.ndarray) -> np.ndarray:
        gray = np.ascontiguousarray(gray, dtype=np.uint8)
        out = np.empty_like(gray)
        self.lib.orc_highpass3(_p(gray), gray.shape[0], gray.shape[1], _p(out))
        return out

    def gauss5(self, gray: np.ndarray) -> np.ndarray:
        gray = np.ascontiguousarray(gray, dtype=np.uint8)
        out = np.empty_like(gray)
        self.lib.orc_gauss5(_p(gray), gray.shape[0], gray.shape[1], _p(out))
        return out

    def quality_sums(self, gray: np.ndarray):
        """(sum, sum of squares) of the 3x3 Laplacian and the sum of squared gray levels (main.py:209-210)."""
        gray = np.ascontiguousarray(gray, dtype=np.uint8)
        s, q, g2 = C.c_int64(), C.c_uint64(), C.c_uint64()
        self.lib.orc_quality_sums(_p(gray), gray.shape[0], gray.shape[1], C.byref(s), C.byref(q), C.byref(g2))
        return int(s.value), int(q.value), int(g2.value)

    def canny(self, gray: np.ndarray, low: int = 50, high: int = 150):
        gray = np.ascontiguousarray(gray, dtype=np.uint8)
        edges = np.empty_like(gray)
        nms = np.empty_like(gray)
        cnt = self.lib.orc_canny(_p(gray), gray.shape[0], gray.shape[1], low, high, _p(edges), _p(nms))
        if cnt < 0:
            raise MemoryError
        return edges, nms, cnt

    def fft_stats(self, gray: np.ndarray, want_spectrum: bool = False):
        gray = np.ascontiguousarray(gray, dtype=np.uint8)
        h, w = gray.shape
        s = C.c_double()
        s2 = C.c_double()
        spec = np.empty((h, w, 2), np.float64) if want_spectrum else None
        sp = spec.ctypes.data_as(C.POINTER(C.c_double)) if want_spectrum else None
        if self.lib.orc_fft_logmag_stats(_p(gray), h, w, w, C.byref(s), C.byref(s2), sp) != 0:
            raise MemoryError
        return s.value, s2.value, (spec[..., 0] + 1j * spec[..., 1] if want_spectrum else None)

    def analyze_frame(self, bgr: np.ndarray, params: OrcParams | None = None) -> OrcMetrics:
        bgr = np.ascontiguousarray(bgr, dtype=np.uint8)
        h, w, c = bgr.shape
        assert c == 3
        m = OrcMetrics()
        rc = self.lib.orc_analyze_frame(_p(bgr), h, w, w * 3, C.byref(params) if params is not None else None, C.byref(m))
        if rc != 0:
            raise ValueError("orc_analyze_frame failed")
        return m

    def synth_frame(self, kind: int, seed: int, video: int, frame: int, h: int, w: int) -> np.ndarray:
        out = np.empty((h, w, 3), np.uint8)
        if self.lib.orc_synth_frame(kind, seed, video, frame, h, w, _p(out)) != 0:
            raise ValueError("bad synth arguments")
        return out


METRIC_KEYS = ("lbp_variance", "noise_level", "compression_artifacts", "edge_density", "frequency_anomaly")


def metrics_to_dict(m: OrcMetrics) -> dict:
    return {k: float(getattr(m, k)) for k in METRIC_KEYS}


# --------------------------------------------------------------------------- numpy / scipy mirror


def np_gray(bgr: np.ndarray, shift: int = 15) -> np.ndarray:
    """cv2.cvtColor(BGR2GRAY), passive_checker.py:100 (fixed point, round to nearest)."""
    b = bgr[..., 0].astype(np.int64)
    g = bgr[..., 1].astype(np.int64)
    r = bgr[..., 2].astype(np.int64)
    if shift == 15:
        return ((b * 3735 + g * 19235 + r * 9798 + 16384) >> 15).astype(np.uint8)
    return ((b * 1868 + g * 9617 + r * 4899 + 8192) >> 14).astype(np.uint8)


def _pad101(a: np.ndarray, n: int) -> np.ndarray:
    """BORDER_REFLECT_101 padding that also works when a side is shorter than n+1."""
    def idx(length):
        i = np.arange(-n, length + n)
        if length == 1:
            return np.zeros_like(i)
        period = 2 * length - 2
        i = np.mod(i, period)
        return np.where(i >= length, period - i, i)
    return a[np.ix_(idx(a.shape[0]), idx(a.shape[1]))]


def np_highpass(gray: np.ndarray) -> np.ndarray:
    """cv2.filter2D(gray, -1, 3x3 high-pass), passive_checker.py:107."""
    from scipy import ndimage
    k = np.array([[-1, -1, -1], [-1, 8, -1], [-1, -1, -1]], dtype=np.int64)
    p = _pad101(gray.astype(np.int64), 1)
    full = ndimage.correlate(p, k, mode="constant", cval=0)
    return np.clip(full[1:-1, 1:-1], 0, 255).astype(np.uint8)


def np_gauss5(gray: np.ndarray) -> np.ndarray:
    """cv2.GaussianBlur(gray, (5,5), 0), passive_checker.py:111."""
    w = np.array([1, 4, 6, 4, 1], dtype=np.int64)
    p = _pad101(gray.astype(np.int64), 2)
    h_, w_ = gray.shape
    hor = sum(w[i] * p[:, i:i + w_] for i in range(5))
    ver = sum(w[i] * hor[i:i + h_, :] for i in range(5))
    return ((ver + 128) >> 8).astype(np.uint8)


def np_canny(gray: np.ndarray, low: int = 50, high: int = 150):
    """cv2.Canny(gray, 50, 150), passive_checker.py:116.  Returns (edges u8 0/255, nms map 0/1/2)."""
    from scipy import ndimage
    h, w = gray.shape
    if low > high:  # cv2.Canny swaps reversed thresholds
        low, high = high, low
    g = np.pad(gray.astype(np.int64), 1, mode="edge")  # BORDER_REPLICATE
    gx = (g[:-2, 2:] - g[:-2, :-2]) + 2 * (g[1:-1, 2:] - g[1:-1, :-2]) + (g[2:, 2:] - g[2:, :-2])
    gy = (g[2:, :-2] - g[:-2, :-2]) + 2 * (g[2:, 1:-1] - g[:-2, 1:-1]) + (g[2:, 2:] - g[:-2, 2:])
    mag = np.abs(gx) + np.abs(gy)
    mp = np.pad(mag, 1, mode="constant")

    def nb(dy, dx):
        return mp[1 + dy:1 + dy + h, 1 + dx:1 + dx + w]

    ax = np.abs(gx)
    ay = np.abs(gy) << 15
    tg22 = ax * 13573
    tg67 = tg22 + (ax << 16)
    horiz = ay < tg22
    vert = (~horiz) & (ay > tg67)
    diag = ~(horiz | vert)
    opposite = (gx ^ gy) < 0
    keep_h = (mag > nb(0, -1)) & (mag >= nb(0, 1))
    keep_v = (mag > nb(-1, 0)) & (mag >= nb(1, 0))
    keep_d_same = (mag > nb(-1, -1)) & (mag > nb(1, 1))
    keep_d_opp = (mag > nb(-1, 1)) & (mag > nb(1, -1))
    keep = (horiz & keep_h) | (vert & keep_v) | (diag & np.where(opposite, keep_d_opp, keep_d_same))
    keep &= mag > low
    nms = np.where(keep, np.where(mag > high, 2, 1), 0).astype(np.uint8)
    lab, n = ndimage.label(nms > 0, structure=np.ones((3, 3), dtype=int))
    if n:
        has_strong = ndimage.maximum(nms, lab, index=np.arange(1, n + 1)) == 2
        ok = np.concatenate([[False], has_strong])
        edges = np.where(ok[lab], 255, 0).astype(np.uint8)
    else:
        edges = np.zeros_like(nms)
    return edges, nms


def np_lbp_uniform(gray: np.ndarray, P: int = 8, R: float = 1.0) -> np.ndarray:
    """skimage.feature.local_binary_pattern(gray, P, R, 'uniform'), passive_checker.py:103.

    Vectorised over the image; every fp64 operation is the same single
    numpy ufunc call the scalar algorithm performs, in the same order.
    """
    h, w = gray.shape
    img = gray.astype(np.float64)
    ang = 2 * np.pi * np.arange(P, dtype=np.double) / P
    rp = np.round(-R * np.sin(ang), 5)
    cp = np.round(R * np.cos(ang), 5)
    rr = np.arange(h, dtype=np.float64)[:, None]
    cc = np.arange(w, dtype=np.float64)[None, :]
    pad = int(np.ceil(R)) + 1
    z = np.zeros((h + 2 * pad, w + 2 * pad), np.float64)
    z[pad:pad + h, pad:pad + w] = img

    def at(ri, ci):
        return z[ri + pad, ci + pad]

    bits = np.empty((P, h, w), np.int8)
    for i in range(P):
        rf = rr + rp[i]
        cf = cc + cp[i]
        minr = np.floor(rf); maxr = np.ceil(rf)
        minc = np.floor(cf); maxc = np.ceil(cf)
        dr = rf - minr
        dc = cf - minc
        minr_i = np.broadcast_to(minr.astype(np.int64), (h, w)); maxr_i = np.broadcast_to(maxr.astype(np.int64), (h, w))
        minc_i = np.broadcast_to(minc.astype(np.int64), (h, w)); maxc_i = np.broadcast_to(maxc.astype(np.int64), (h, w))
        tl = at(minr_i, minc_i); tr = at(minr_i, maxc_i)
        bl = at(maxr_i, minc_i); br = at(maxr_i, maxc_i)
        top = (1 - dc) * tl + dc * tr
        bottom = (1 - dc) * bl + dc * br
        t = (1 - dr) * top + dr * bottom
        bits[i] = (t - img) >= 0
    changes = (bits[:-1] != bits[1:]).sum(axis=0)
    ones = bits.sum(axis=0)
    return np.where(changes <= 2, ones, P + 1).astype(np.uint8)


def np_analyze_frame(bgr: np.ndarray, P: int = 8, R: float = 1.0, low: int = 50, high: int = 150,
                     shift: int = 15) -> dict:
    """The five metrics of passive_checker.py:96-131 with numpy doing every reduction the reference gives it."""
    gray = np_gray(bgr, shift)
    lbp = np_lbp_uniform(gray, P, R).astype(np.float64)
    hf = np_highpass(gray)
    blur = np_gauss5(gray)
    diff = np.abs(gray.astype(np.int16) - blur.astype(np.int16)).astype(np.uint8)
    edges, _ = np_canny(gray, low, high)
    mag = np.log(np.abs(np.fft.fftshift(np.fft.fft2(gray))) + 1)
    return {
        "lbp_variance": float(np.var(lbp)),
        "noise_level": float(np.std(hf)),
        "compression_artifacts": float(np.mean(diff)),
        "edge_density": float(np.sum(edges > 0) / edges.size),
        "frequency_anomaly": float(np.std(mag)),
    }


def np_texture_score(m: dict) -> float:
    """passive_checker.py:145-151."""
    return (
        min(m["lbp_variance"] / 50.0, 1.0) * 0.3
        + min(m["edge_density"] * 10, 1.0) * 0.2
        + (1.0 - min(m["compression_artifacts"] / 10.0, 1.0)) * 0.2
        + min(m["noise_level"] / 20.0, 1.0) * 0.15
        + (1.0 - min(m["frequency_anomaly"] / 100.0, 1.0)) * 0.15
    )


def np_compute_texture_score(frames, threshold: float = 0.6, **kw) -> dict:
    """passive_checker.py:133-164."""
    scores, detailed = [], []
    for frame in frames[:10]:
        m = np_analyze_frame(frame, **kw)
        detailed.append(m)
        scores.append(np_texture_score(m))
    avg = np.mean(scores) if scores else 0.0
    passed = avg >= threshold
    return {
        "passed": bool(passed),
        "average_score": round(float(avg), 3),
        "individual_scores": [round(float(s), 3) for s in scores],
        "detailed_metrics": detailed,
        "message": f"Texture analysis {'passed' if passed else 'failed'} (score: {avg:.3f})",
    }


# ---------------------------------------------------------------------------- callers of the path
def np_quality_sums(gray: np.ndarray):
    """Independent restatement of the fallback checkers' per-frame inputs (main.py:209-210): the image
    cv2.Laplacian(gray, CV_64F) (aperture 1: [0 1 0; 1 -4 1; 0 1 0], BORDER_REFLECT_101) as exact integers."""
    from scipy import ndimage
    k = np.array([[0, 1, 0], [1, -4, 1], [0, 1, 0]], dtype=np.int64)
    lap = ndimage.correlate(gray.astype(np.int64), k, mode="mirror")
    g = gray.astype(np.int64)
    return int(lap.sum()), int((lap * lap).sum()), int((g * g).sum())


def np_quality_variances(gray: np.ndarray):
    """(Laplacian variance, gray variance) the way the reference gets them: numpy's .var() on the images."""
    from scipy import ndimage
    k = np.array([[0, 1, 0], [1, -4, 1], [0, 1, 0]], dtype=np.float64)
    lap = ndimage.correlate(gray.astype(np.float64), k, mode="mirror")
    return float(lap.var()), float(gray.var())


def np_fallback_check(frames, threshold: float = 0.5) -> dict:
    """main.py:105-141 on the numpy restatement."""
    if not frames:
        return {"passed": False, "average_real_score": 0.0, "message": "No frames to analyze"}
    scores = []
    for f in frames[:10]:
        lv, bv = np_quality_variances(np_gray(np.asarray(f)))
        scores.append((min(lv / 1000.0, 1.0) + min(bv / 10000.0, 1.0)) / 2.0)
    avg = sum(scores) / len(scores) if scores else 0.5
    return {"passed": bool(avg >= threshold), "average_real_score": round(float(avg), 3),
            "message": "Fallback passive check (basic image quality analysis)"}


def np_improved_fallback_base(frames) -> float:
    """Base score of main.py:188-250 with no face detector: first fifteen frames."""
    scores = []
    for f in frames[:15]:
        g = np_gray(np.asarray(f))
        lv, bv = np_quality_variances(g)
        edges, _ = np_canny(g, 50, 150)
        ed = np.count_nonzero(edges) / edges.size
        scores.append(min(lv / 500.0, 1.0) * 0.4 + min(bv / 5000.0, 1.0) * 0.3 + min(ed * 20, 1.0) * 0.3)
    return min(sum(scores) / len(scores), 1.0)


# ---------------------------------------------------------------------------- frame preparation (CNN branch)
def _resize_taps(src: int, dst: int, horizontal: bool):
    """Tap positions and 11-bit weights of cv::resize(INTER_LINEAR) along one axis (8-bit images).
    coordinate = (d + 0.5) * scale - 0.5 rounded to float32; weights saturate_cast<short>(w * 2048);
    columns outside the image collapse onto the border column with weight 2048, rows are clamped and
    keep their weights."""
    scale = 1.0 / (float(dst) / float(src))
    i0 = np.zeros(dst, np.int64)
    i1 = np.zeros(dst, np.int64)
    w0 = np.zeros(dst, np.int64)
    w1 = np.zeros(dst, np.int64)
    for d in range(dst):
        f = np.float32((d + 0.5) * scale - 0.5)
        s = int(np.floor(f))
        f = np.float32(f - np.float32(s))
        if horizontal:
            if s < 0:
                f, s = np.float32(0), 0
            if s + 1 >= src:
                f, s = np.float32(0), src - 1
                t = s
            else:
                t = s + 1
        else:
            t = min(max(s + 1, 0), src - 1)
            s = min(max(s, 0), src - 1)
        i0[d], i1[d] = s, t
        w0[d] = int(np.rint(np.float32(np.float32(1) - f) * np.float32(2048)))
        w1[d] = int(np.rint(f * np.float32(2048)))
    return i0, i1, w0, w1


def np_resize_linear_u8(img: np.ndarray, out_h: int, out_w: int) -> np.ndarray:
    """cv2.resize(img, (out_w, out_h)) for uint8 H x W x C images, default interpolation
    (passive_checker.py:407).  Restated from OpenCV's algorithm; no OpenCV in this image to pin it."""
    h, w = img.shape[:2]
    a = img.astype(np.int64)
    sx, sy = 1.0 / (float(out_w) / w), 1.0 / (float(out_h) / h)
    eps = np.finfo(np.float64).eps
    if abs(sx - 2.0) < eps and abs(sy - 2.0) < eps:
        # exact 2:1 reduction: OpenCV switches to the 2x2 area average
        s = a[0:2 * out_h:2, 0:2 * out_w:2] + a[0:2 * out_h:2, 1:2 * out_w:2] + \
            a[1:2 * out_h:2, 0:2 * out_w:2] + a[1:2 * out_h:2, 1:2 * out_w:2]
        return ((s + 2) >> 2).astype(np.uint8)
    xi0, xi1, xw0, xw1 = _resize_taps(w, out_w, True)
    yi0, yi1, yw0, yw1 = _resize_taps(h, out_h, False)
    shp = (1, out_w) + (1,) * (a.ndim - 2)
    rows = a[:, xi0] * xw0.reshape(shp) + a[:, xi1] * xw1.reshape(shp)  # [h][out_w][C]
    shp = (out_h, 1) + (1,) * (a.ndim - 2)
    s0, s1 = rows[yi0], rows[yi1]
    v = (((yw0.reshape(shp) * (s0 >> 4)) >> 16) + ((yw1.reshape(shp) * (s1 >> 4)) >> 16) + 2) >> 2
    return np.clip(v, 0, 255).astype(np.uint8)


def np_preprocess_frame(frame: np.ndarray, input_size=(80, 80)) -> np.ndarray:
    """passive_checker.py:393-418 without the torch wrapper: [1][3][H][W] float32."""
    rgb = np.asarray(frame)[:, :, ::-1]
    resized = np_resize_linear_u8(rgb, input_size[1], input_size[0])
    normalized = resized.astype(np.float32) / 255.0
    return np.transpose(normalized, (2, 0, 1))[None]


# ---------------------------------------------------------------------------- extended statistics
def np_extended_sums(gray: np.ndarray, high: int = 150) -> dict:
    """Exact integer sums behind the extended statistics (no counterpart in the reference; the definitions are
    this project's own): Sobel L1 magnitude with replicated borders -- the images cv2.Canny builds at
    passive_checker.py:116 -- and absolute neighbour differences split into 8-pixel block boundaries and block
    interiors, horizontally and vertically."""
    from scipy import ndimage
    g = gray.astype(np.int64)
    kx = np.array([[-1, 0, 1], [-2, 0, 2], [-1, 0, 1]], dtype=np.int64)
    gx = ndimage.correlate(g, kx, mode="nearest")
    gy = ndimage.correlate(g, kx.T, mode="nearest")
    mag = np.abs(gx) + np.abs(gy)
    h, w = g.shape
    dh = np.abs(g[:, 1:] - g[:, :-1])  # column x = 1..w-1
    dv = np.abs(g[1:, :] - g[:-1, :])
    xb = (np.arange(1, w) % 8) == 0
    yb = (np.arange(1, h) % 8) == 0
    return {
        "sobel_sum": int(mag.sum()), "sobel_sumsq": int((mag * mag).sum()), "sobel_strong": int((mag > high).sum()),
        "block_sums": [int(dh[:, xb].sum()), int(dh[:, ~xb].sum()), int(dv[yb, :].sum()), int(dv[~yb, :].sum())],
        "block_counts": [int(h * xb.sum()), int(h * (~xb).sum()), int(w * yb.sum()), int(w * (~yb).sum())],
    }


def np_extended_stats(gray: np.ndarray, edge_count: int, high: int = 150) -> dict:
    s = np_extended_sums(gray, high)
    n = gray.size
    mean = s["sobel_sum"] / n
    var = (n * s["sobel_sumsq"] - s["sobel_sum"] ** 2) / (n * n)
    ne = s["block_counts"][0] + s["block_counts"][2]
    ni = s["block_counts"][1] + s["block_counts"][3]
    me = (s["block_sums"][0] + s["block_sums"][2]) / ne if ne else 0.0
    mi = (s["block_sums"][1] + s["block_sums"][3]) / ni if ni else 0.0
    s.update(sobel_mean=mean, sobel_std=float(np.sqrt(var)), strong_gradient_fraction=s["sobel_strong"] / n,
             edge_consistency=(edge_count / s["sobel_strong"]) if s["sobel_strong"] else 0.0,
             blockiness=me / (mi + 1e-10))
    return s
