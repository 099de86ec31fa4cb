"""ctypes binding of libsift_texture.so (declared in include/sift_texture.h).

There is no CPU fallback: if the shared library is missing or no GPU is usable the
import of the library or the creation of a context raises.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# SIFT_TEXTURE_LIB selects another build of the library (A/B timing of kernels)
LIB_PATH = os.environ.get("SIFT_TEXTURE_LIB") or os.path.join(_HERE, "libsift_texture.so")

STX_OK = 0
FLAG_FFT_F64 = 1
LBP_MAX_BINS = 26
N_KERNELS = 10
KERNEL_NAMES = ("gray", "stencil", "lbp_generic", "hysteresis", "fft_rows", "fft_cols", "reduce", "synth", "quality",
                "extended")

EXPORTS = (
    "stx_version", "stx_create", "stx_destroy", "stx_last_error", "stx_backend", "stx_default_params",
    "stx_analyze_batch", "stx_analyze_frames", "stx_analyze_batch_device", "stx_analyze_batch_device_noalloc", "stx_submit_batch_device", "stx_wait",
    "stx_max_async_frames", "stx_score_summary",
    "stx_debug_planes", "stx_synth_device", "stx_device_alloc", "stx_device_free", "stx_pinned_alloc",
    "stx_pinned_free", "stx_memcpy_d2h", "stx_memcpy_h2d", "stx_set_chunk_frames", "stx_set_concurrency", "stx_set_timing", "stx_set_canny_mode",
    "stx_get_timing", "stx_preprocess_frames", "stx_preprocess_batch_device",
    "stx_analyze_extended_frames", "stx_analyze_extended_batch_device", "stx_submit_extended_batch_device",
    "stx_synth_device_async",
)


class StxError(RuntimeError):
    pass


class StxParams(C.Structure):
    _fields_ = [
        ("lbp_points", C.c_int),
        ("lbp_radius", C.c_double),
        ("canny_low", C.c_int),
        ("canny_high", C.c_int),
        ("gray_shift", C.c_int),
        ("fft_h", C.c_int),
        ("fft_w", C.c_int),
        ("quality_stats", C.c_int),
        ("fft_exact", C.c_int),
    ]


class StxMetrics(C.Structure):
    _fields_ = [
        ("lbp_variance", C.c_double),
        ("noise_level", C.c_double),
        ("compression_artifacts", C.c_double),
        ("edge_density", C.c_double),
        ("frequency_anomaly", C.c_double),
        ("texture_score", C.c_double),
        ("lbp_hist", C.c_uint32 * LBP_MAX_BINS),
        ("edge_count", C.c_uint32),
        ("flags", C.c_uint32),
        ("absdiff_sum", C.c_uint64),
        ("hp_sum", C.c_uint64),
        ("hp_sumsq", C.c_uint64),
        ("gray_sum", C.c_uint64),
        ("n_pixels", C.c_uint64),
        ("fft_sum", C.c_double),
        ("fft_sumsq", C.c_double),
        ("laplacian_variance", C.c_double),
        ("brightness_variance", C.c_double),
        ("lap_sum", C.c_int64),
        ("lap_sumsq", C.c_uint64),
        ("gray_sumsq", C.c_uint64),
    ]


class StxExtended(C.Structure):
    _fields_ = [
        ("sobel_mean", C.c_double),
        ("sobel_std", C.c_double),
        ("strong_gradient_fraction", C.c_double),
        ("edge_consistency", C.c_double),
        ("blockiness", C.c_double),
        ("sobel_sum", C.c_uint64),
        ("sobel_sumsq", C.c_uint64),
        ("block_sums", C.c_uint64 * 4),
        ("block_counts", C.c_uint64 * 4),
        ("sobel_strong", C.c_uint32),
        ("reserved", C.c_uint32),
        ("lbp_hist_p16r2", C.c_uint32 * LBP_MAX_BINS),
        ("lbp_hist_p24r3", C.c_uint32 * LBP_MAX_BINS),
    ]


class StxTiming(C.Structure):
    _fields_ = [("ms", C.c_double * N_KERNELS), ("launches", C.c_uint64 * N_KERNELS), ("frames", C.c_uint64),
                ("fft64_frames", C.c_uint64)]


_lib = None


def load() -> C.CDLL:
    """Load the shared library once; raise if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise StxError(
            f"{LIB_PATH} is missing: build it with `make -C sift_amd/csrc` (or __graft_entry__.build()); "
            "there is no CPU fallback")
    # PyTorch's wheels carry their own copy of the HIP runtime.  A process can hold the GPU through one
    # runtime only: if this library pulled in the system's copy first, torch.cuda would later find no device.
    # Importing torch first makes both use the copy torch ships.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    u8p = C.c_void_p
    L.stx_version.restype = C.c_char_p
    L.stx_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
    L.stx_destroy.argtypes = [C.c_void_p]
    L.stx_last_error.argtypes = [C.c_void_p]
    L.stx_last_error.restype = C.c_char_p
    L.stx_backend.argtypes = [C.c_void_p]
    L.stx_default_params.argtypes = [C.POINTER(StxParams)]
    L.stx_default_params.restype = None
    L.stx_analyze_batch.argtypes = [C.c_void_p, u8p, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int64,
                                    C.POINTER(StxParams), C.POINTER(StxMetrics)]
    L.stx_analyze_frames.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_int64,
                                     C.POINTER(StxParams), C.POINTER(StxMetrics)]
    L.stx_analyze_batch_device.argtypes = L.stx_analyze_batch.argtypes
    L.stx_analyze_batch_device_noalloc.argtypes = L.stx_analyze_batch.argtypes
    L.stx_submit_batch_device.argtypes = L.stx_analyze_batch.argtypes + [C.POINTER(C.c_int)]
    L.stx_wait.argtypes = [C.c_void_p, C.c_int]
    L.stx_max_async_frames.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.stx_score_summary.argtypes = [C.POINTER(StxMetrics), C.c_int, C.c_double, C.POINTER(C.c_double),
                                    C.POINTER(C.c_int)]
    L.stx_debug_planes.argtypes = [C.c_void_p, u8p, C.c_int, C.c_int, C.c_int64, C.POINTER(StxParams), u8p, u8p, u8p]
    L.stx_synth_device.argtypes = [C.c_void_p, u8p, C.c_int, C.c_uint64, C.c_uint32, C.c_uint32, C.c_int, C.c_int,
                                   C.c_int]
    L.stx_synth_device_async.argtypes = L.stx_synth_device.argtypes
    L.stx_device_alloc.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(C.c_void_p)]
    L.stx_device_free.argtypes = [C.c_void_p, C.c_void_p]
    L.stx_pinned_alloc.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(C.c_void_p)]
    L.stx_pinned_free.argtypes = [C.c_void_p, C.c_void_p]
    L.stx_memcpy_d2h.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]
    L.stx_memcpy_h2d.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]
    L.stx_set_chunk_frames.argtypes = [C.c_void_p, C.c_int]
    L.stx_set_concurrency.argtypes = [C.c_void_p, C.c_int]
    L.stx_set_timing.argtypes = [C.c_void_p, C.c_int]
    L.stx_set_canny_mode.argtypes = [C.c_void_p, C.c_int]
    L.stx_get_timing.argtypes = [C.c_void_p, C.POINTER(StxTiming), C.c_int]
    L.stx_preprocess_frames.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_int64,
                                        C.c_int, C.c_int, C.c_void_p]
    L.stx_preprocess_batch_device.argtypes = [C.c_void_p, u8p, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int64,
                                              C.c_int, C.c_int, C.c_void_p]
    L.stx_analyze_extended_frames.argtypes = L.stx_analyze_frames.argtypes + [C.POINTER(StxExtended)]
    L.stx_analyze_extended_batch_device.argtypes = L.stx_analyze_batch.argtypes + [C.POINTER(StxExtended)]
    L.stx_submit_extended_batch_device.argtypes = L.stx_analyze_extended_batch_device.argtypes + [C.POINTER(C.c_int)]
    _lib = L
    return L
