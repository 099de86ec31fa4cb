"""Drop-in for the reference's texture/noise detector.

``TextureNoiseDetector`` mirrors /root/reference/app/passive_checker.py:88-164: same
constructor, ``threshold`` attribute, method names, argument meaning, returned dict keys
(in the same insertion order) and JSON-native leaf types.  The per-frame arithmetic runs
in libsift_texture.so (HIP, gfx950); this module only moves frames across the C ABI and
formats the result.  Failures surface as ordinary Python exceptions.
"""
from __future__ import annotations

import ctypes as C
from typing import Any, Dict, List, Optional, Sequence

import numpy as np

from . import _lib
from ._lib import StxError, StxExtended, StxMetrics, StxParams, StxTiming

METRIC_KEYS = ("lbp_variance", "noise_level", "compression_artifacts", "edge_density", "frequency_anomaly")


class TextureContext:
    """One GPU context (stream + workspace) of the C library."""

    def __init__(self, device: int = 0):
        self.lib = _lib.load()
        h = C.c_void_p()
        rc = self.lib.stx_create(int(device), C.byref(h))
        if rc != _lib.STX_OK or not h:
            why = self.lib.stx_last_error(None)
            raise StxError(f"stx_create(device={device}) failed with code {rc} "
                           f"({why.decode() if why else 'unknown'}): no usable MI355X / HIP runtime")
        self.handle = h
        self.device = int(device)

    def close(self):
        if getattr(self, "handle", None):
            self.lib.stx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int, what: str):
        if rc != _lib.STX_OK:
            msg = self.lib.stx_last_error(self.handle)
            raise StxError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")

    def default_params(self) -> StxParams:
        p = StxParams()
        self.lib.stx_default_params(C.byref(p))
        return p

    @staticmethod
    def _frame_ptrs(frames: Sequence[np.ndarray]):
        """Pointers to n same-shaped uint8 BGR frames with one common row stride (copies those that differ)."""
        n = len(frames)
        h, w, ch = frames[0].shape
        if ch != 3:
            raise ValueError("frames must be H x W x 3 (BGR)")
        keep = []
        for f in frames:
            if f.shape != (h, w, 3):
                raise ValueError("all frames of one call must have the same shape")
            if f.dtype != np.uint8 or f.strides[2] != 1 or f.strides[1] != 3 or f.strides[0] < 3 * w:
                f = np.ascontiguousarray(f, dtype=np.uint8)
            keep.append(f)
        # the call takes one row stride: views with padded rows pass as they are when all of them agree, any
        # mix of strides (a crop view beside a contiguous frame) is made contiguous first
        if len({f.strides[0] for f in keep}) > 1:
            keep = [np.ascontiguousarray(f) for f in keep]
        ptrs = (C.c_void_p * n)()
        for i, f in enumerate(keep):
            ptrs[i] = f.ctypes.data
        return keep, ptrs, h, w, keep[0].strides[0]

    # ---- host frames
    def analyze_frames(self, frames: Sequence[np.ndarray], params: Optional[StxParams] = None):
        """All frames must share one shape [H][W][3] uint8.  Returns a ctypes array of StxMetrics."""
        n = len(frames)
        out = (StxMetrics * max(n, 1))()
        if n == 0:
            return out
        keep, ptrs, h, w, stride = self._frame_ptrs(frames)
        rc = self.lib.stx_analyze_frames(self.handle, ptrs, n, h, w, stride, C.byref(params) if params is not None else None, out)
        self._check(rc, "stx_analyze_frames")
        return out

    def analyze_batch(self, batch: np.ndarray, params: Optional[StxParams] = None):
        """batch: [N][H][W][3] uint8 C-contiguous host array."""
        batch = np.ascontiguousarray(batch, dtype=np.uint8)
        n, h, w, ch = batch.shape
        if ch != 3:
            raise ValueError("batch must be N x H x W x 3 (BGR)")
        out = (StxMetrics * max(n, 1))()
        rc = self.lib.stx_analyze_batch(self.handle, batch.ctypes.data, n, h, w, 3 * w, 3 * w * h,
                                        C.byref(params) if params is not None else None, out)
        self._check(rc, "stx_analyze_batch")
        return out

    # ---- device frames
    def analyze_device(self, dev_ptr: int, n: int, h: int, w: int, params: Optional[StxParams] = None,
                       row_stride: Optional[int] = None, frame_stride: Optional[int] = None, out=None):
        if out is None:
            out = (StxMetrics * max(n, 1))()
        rs = 3 * w if row_stride is None else row_stride
        fs = rs * h if frame_stride is None else frame_stride
        rc = self.lib.stx_analyze_batch_device(self.handle, dev_ptr, n, h, w, rs, fs,
                                               C.byref(params) if params is not None else None, out)
        self._check(rc, "stx_analyze_batch_device")
        return out

    def submit_device(self, dev_ptr: int, n: int, h: int, w: int, out, params: Optional[StxParams] = None) -> int:
        """Queue the analysis of n device-resident frames; returns a ticket for wait().  `out` (a ctypes
        array of StxMetrics) and `params` must stay alive until wait() returns."""
        t = C.c_int(-1)
        rc = self.lib.stx_submit_batch_device(self.handle, dev_ptr, n, h, w, 3 * w, 3 * w * h,
                                              C.byref(params) if params is not None else None, out, C.byref(t))
        self._check(rc, "stx_submit_batch_device")
        return t.value

    def submit_device_extended(self, dev_ptr: int, n: int, h: int, w: int, out, ext,
                               params: Optional[StxParams] = None) -> int:
        """submit_device plus one StxExtended record per frame in `ext`."""
        t = C.c_int(-1)
        rc = self.lib.stx_submit_extended_batch_device(self.handle, dev_ptr, n, h, w, 3 * w, 3 * w * h,
                                                       C.byref(params) if params is not None else None, out, ext,
                                                       C.byref(t))
        self._check(rc, "stx_submit_extended_batch_device")
        return t.value

    def wait(self, ticket: int):
        self._check(self.lib.stx_wait(self.handle, ticket), "stx_wait")

    def max_async_frames(self, h: int, w: int) -> int:
        return int(self.lib.stx_max_async_frames(self.handle, h, w))

    # ---- extended statistics
    def analyze_frames_extended(self, frames: Sequence[np.ndarray], params: Optional[StxParams] = None):
        """analyze_frames plus one StxExtended record per frame: (metrics array, extended array)."""
        n = len(frames)
        out = (StxMetrics * max(n, 1))()
        ext = (StxExtended * max(n, 1))()
        if n == 0:
            return out, ext
        keep, ptrs, h, w, stride = self._frame_ptrs(frames)
        rc = self.lib.stx_analyze_extended_frames(self.handle, ptrs, n, h, w, stride,
                                                  C.byref(params) if params is not None else None, out, ext)
        self._check(rc, "stx_analyze_extended_frames")
        return out, ext

    def analyze_device_extended(self, dev_ptr: int, n: int, h: int, w: int, params: Optional[StxParams] = None):
        out = (StxMetrics * max(n, 1))()
        ext = (StxExtended * max(n, 1))()
        rc = self.lib.stx_analyze_extended_batch_device(self.handle, dev_ptr, n, h, w, 3 * w, 3 * w * h,
                                                        C.byref(params) if params is not None else None, out, ext)
        self._check(rc, "stx_analyze_extended_batch_device")
        return out, ext

    # ---- frame preparation of the CNN branch
    def preprocess_frames(self, frames: Sequence[np.ndarray], out_h: int = 80, out_w: int = 80) -> np.ndarray:
        """[N][3][out_h][out_w] float32 (RGB planes in 0..1) from N same-shaped BGR frames."""
        n = len(frames)
        out = np.empty((n, 3, out_h, out_w), np.float32)
        if n == 0:
            return out
        keep, ptrs, h, w, stride = self._frame_ptrs(frames)
        rc = self.lib.stx_preprocess_frames(self.handle, ptrs, n, h, w, stride, out_h, out_w, out.ctypes.data)
        self._check(rc, "stx_preprocess_frames")
        return out

    def preprocess_device(self, dev_ptr: int, n: int, h: int, w: int, out_ptr: int, out_h: int = 80, out_w: int = 80,
                          row_stride: Optional[int] = None, frame_stride: Optional[int] = None):
        rs = 3 * w if row_stride is None else row_stride
        fs = rs * h if frame_stride is None else frame_stride
        rc = self.lib.stx_preprocess_batch_device(self.handle, dev_ptr, n, h, w, rs, fs, out_h, out_w, out_ptr)
        self._check(rc, "stx_preprocess_batch_device")

    def synth_device(self, dev_ptr: int, kind: int, seed: int, video: int, frame0: int, n: int, h: int, w: int):
        self._check(self.lib.stx_synth_device(self.handle, dev_ptr, kind, seed, video, frame0, n, h, w), "stx_synth_device")

    def synth_device_async(self, dev_ptr: int, kind: int, seed: int, video: int, frame0: int, n: int, h: int, w: int):
        """synth_device without the wait: ordered before this context's later analysis calls."""
        self._check(self.lib.stx_synth_device_async(self.handle, dev_ptr, kind, seed, video, frame0, n, h, w),
                    "stx_synth_device_async")

    def device_alloc(self, nbytes: int) -> int:
        p = C.c_void_p()
        self._check(self.lib.stx_device_alloc(self.handle, nbytes, C.byref(p)), "stx_device_alloc")
        return p.value

    def device_free(self, ptr: int):
        self._check(self.lib.stx_device_free(self.handle, ptr), "stx_device_free")

    def d2h(self, dev_ptr: int, nbytes: int) -> np.ndarray:
        out = np.empty(nbytes, np.uint8)
        self._check(self.lib.stx_memcpy_d2h(self.handle, out.ctypes.data, dev_ptr, nbytes), "stx_memcpy_d2h")
        return out

    def h2d(self, dev_ptr: int, arr: np.ndarray):
        arr = np.ascontiguousarray(arr)
        self._check(self.lib.stx_memcpy_h2d(self.handle, dev_ptr, arr.ctypes.data, arr.nbytes), "stx_memcpy_h2d")

    def debug_planes(self, frame: np.ndarray, params: Optional[StxParams] = None):
        frame = np.ascontiguousarray(frame, dtype=np.uint8)
        h, w, _ = frame.shape
        gray = np.empty((h, w), np.uint8)
        codes = np.empty((h, w), np.uint8)
        edges = np.empty((h, w), np.uint8)
        rc = self.lib.stx_debug_planes(self.handle, frame.ctypes.data, h, w, 3 * w,
                                       C.byref(params) if params is not None else None, gray.ctypes.data,
                                       codes.ctypes.data, edges.ctypes.data)
        self._check(rc, "stx_debug_planes")
        return gray, codes, edges

    def set_chunk_frames(self, n: int):
        self._check(self.lib.stx_set_chunk_frames(self.handle, n), "stx_set_chunk_frames")

    CANNY_MODES = {"dense": 0, "sparse": 1, "auto": 2}

    def set_canny_mode(self, mode: str):
        """Where Canny's non-maximum suppression runs (same edge map either way): "dense" = on every pixel in the
        stencil pass, "sparse" = on the pixels a flood of the raw threshold maps reaches, "auto" (default) = sparse
        while that set stays small."""
        self._check(self.lib.stx_set_canny_mode(self.handle, self.CANNY_MODES[mode]), "stx_set_canny_mode")

    def set_concurrency(self, on: bool):
        self._check(self.lib.stx_set_concurrency(self.handle, 1 if on else 0), "stx_set_concurrency")

    def set_timing(self, on: bool, kinds=None):
        """Per-kernel HIP-event timing on or off; kinds = names of the kernel families to time (default: all)."""
        v = 1 if on else 0
        if on and kinds is not None:
            v = sum(2 << _lib.KERNEL_NAMES.index(k) for k in kinds)
        self._check(self.lib.stx_set_timing(self.handle, v), "stx_set_timing")

    def get_timing(self, reset: bool = False) -> dict:
        t = StxTiming()
        self._check(self.lib.stx_get_timing(self.handle, C.byref(t), 1 if reset else 0), "stx_get_timing")
        return {"frames": int(t.frames), "fft64_frames": int(t.fft64_frames),
                "kernels": {name: {"ms": float(t.ms[i]), "launches": int(t.launches[i])}
                            for i, name in enumerate(_lib.KERNEL_NAMES)}}


def metrics_dict(m: StxMetrics) -> Dict[str, float]:
    """The dict of passive_checker.py:125-131 (same keys, same order, Python floats)."""
    return {k: float(getattr(m, k)) for k in METRIC_KEYS}


class TextureNoiseDetector:
    """Same interface as the reference class (passive_checker.py:88-164), computed on the GPU."""

    def __init__(self, device: int = 0, params: Optional[StxParams] = None):
        self.threshold = 0.6  # passive_checker.py:94
        self._ctx = TextureContext(device)
        self._params = params

    @staticmethod
    def _as_frame(frame) -> np.ndarray:
        a = np.asarray(frame)
        if a.ndim != 3 or a.shape[2] != 3:
            # cv2.cvtColor(BGR2GRAY) rejects anything but 3- or 4-channel input (passive_checker.py:100)
            raise ValueError(f"expected an H x W x 3 BGR frame, got shape {a.shape}")
        if a.dtype != np.uint8:
            raise TypeError(f"expected a uint8 frame, got {a.dtype}")
        if a.shape[0] == 0 or a.shape[1] == 0:
            raise ValueError("empty frame")
        return a

    def analyze_frame(self, frame: np.ndarray) -> Dict[str, float]:
        """passive_checker.py:96-131."""
        a = self._as_frame(frame)
        m = self._ctx.analyze_frames([a], self._params)
        return metrics_dict(m[0])

    def analyze_frames_raw(self, frames: Sequence[np.ndarray]) -> List[StxMetrics]:
        """Per-frame StxMetrics records (metrics + the integer reductions) for frames of any mix of shapes."""
        arrs = [self._as_frame(f) for f in frames]
        out: List[Optional[StxMetrics]] = [None] * len(arrs)
        groups: Dict[tuple, List[int]] = {}
        for i, a in enumerate(arrs):
            groups.setdefault(a.shape, []).append(i)
        for _, idxs in groups.items():
            res = self._ctx.analyze_frames([arrs[i] for i in idxs], self._params)
            for j, i in enumerate(idxs):
                r = StxMetrics()
                C.memmove(C.byref(r), C.byref(res[j]), C.sizeof(StxMetrics))
                out[i] = r
        return out  # type: ignore[return-value]

    def compute_texture_score(self, frames: List[np.ndarray]) -> Dict[str, Any]:
        """passive_checker.py:133-164: the first ten frames, per-frame score, mean, verdict."""
        head = list(frames[:10])
        recs = self.analyze_frames_raw(head) if head else []
        return self.result_from_records(recs)

    def result_from_records(self, recs: Sequence[StxMetrics]) -> Dict[str, Any]:
        """The dict of passive_checker.py:158-164 from per-frame records that a device pass already produced
        (callers that analysed the same frames for their own statistics reuse the pass)."""
        scores = [float(r.texture_score) for r in recs]
        detailed = [metrics_dict(r) for r in recs]
        avg_score = np.mean(scores) if scores else 0.0
        passed = avg_score >= self.threshold
        return {
            "passed": bool(passed),
            "average_score": round(float(avg_score), 3),
            "individual_scores": [round(float(s), 3) for s in scores],
            "detailed_metrics": detailed,
            "message": f"Texture analysis {'passed' if passed else 'failed'} (score: {avg_score:.3f})",
        }
