"""Deterministic synthetic frames for parity tests and benchmarks.

Every pixel is a pure function of (seed, video, frame, y, x): a SplitMix64
hash of a packed 64-bit counter plus integer-only shaping.  The same
arithmetic is implemented in C (oracle/texture_oracle.c, ``orc_synth_frame``)
and in HIP (sift_amd/csrc/stx_synth.hip, ``stx_synth_device``), and all three
produce byte-identical frames; there is no floating point anywhere in the
generator so there is nothing to round differently.

Frame layout is what the reference's frame source hands to the texture path
(``/root/reference/app/video_processor.py:18-49``): ``[H][W][3]`` uint8, BGR
interleaved, C-contiguous.

Kinds
-----
``noise``    i.i.d. uniform bytes per channel (worst case for LBP / Canny).
``natural``  smooth separable sinusoid (period 97 x 71 px, drifting with the
             frame index) + a bright disc of radius H/5 whose centre moves
             (3, 2) px per frame + per-channel noise of sigma ~ 4 + B/G/R
             offsets of -10/0/+10.  Few edges, many near-ties.
``ramp``     gray = (x + 2*y + frame) & 255 on all channels (parity stress).
``flat``     constant 77/150/201 (B/G/R) (every LBP comparison is a tie).
``tie``      gray in {100, 101} chosen by one hash bit (tie-heavy).
``step``     vertical and horizontal step edges of height 90 every 16 px,
             plus hash noise of +-1 (long connected Canny contours).
"""
from __future__ import annotations

import numpy as np

KINDS = ("noise", "natural", "ramp", "flat", "tie", "step")
KIND_ID = {k: i for i, k in enumerate(KINDS)}

_GOLD = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)
_SEEDMUL = np.uint64(0xD1342543DE82EF95)


def _splitmix64(z: np.ndarray) -> np.ndarray:
    z = z + _GOLD
    z = (z ^ (z >> np.uint64(30))) * _M1
    z = (z ^ (z >> np.uint64(27))) * _M2
    return z ^ (z >> np.uint64(31))


def pixel_hash(seed: int, video: int, frame: int, h: int, w: int) -> np.ndarray:
    """uint64 [h][w] hash of the packed counter (video<<36 | frame<<26 | y<<13 | x)."""
    if h > 8192 or w > 8192 or frame >= 1024 or video >= (1 << 28):
        raise ValueError("synthetic generator limits: h,w<=8192, frame<1024, video<2^28")
    y = np.arange(h, dtype=np.uint64)[:, None]
    x = np.arange(w, dtype=np.uint64)[None, :]
    idx = (np.uint64(video) << np.uint64(36)) | (np.uint64(frame) << np.uint64(26)) | (y << np.uint64(13)) | x
    with np.errstate(over="ignore"):
        z = np.array([seed & 0xFFFFFFFFFFFFFFFF], dtype=np.uint64) * _SEEDMUL + idx
        return _splitmix64(z)


def _isin(p: np.ndarray) -> np.ndarray:
    """Integer parabolic sine: phase in 1/65536 turns -> value in [-16384, 16384]."""
    p = p & 0xFFFF
    q = p & 0x7FFF
    v = (q * (32768 - q)) >> 14
    return np.where((p & 0x8000) != 0, -v, v)


def synth_frame(kind: str, seed: int, video: int, frame: int, h: int, w: int) -> np.ndarray:
    """One BGR uint8 frame [h][w][3]."""
    kid = KIND_ID[kind]
    hs = pixel_hash(seed, video, frame, h, w)
    y = np.arange(h, dtype=np.int64)[:, None]
    x = np.arange(w, dtype=np.int64)[None, :]
    out = np.empty((h, w, 3), dtype=np.uint8)
    if kid == 0:  # noise
        for ch in range(3):
            out[..., ch] = ((hs >> np.uint64(8 * ch)) & np.uint64(0xFF)).astype(np.uint8)
        return out
    if kid == 1:  # natural
        sx = _isin((x * 65536) // 97 + frame * 1043)
        sy = _isin((y * 65536) // 71 + 16384)
        base = 128 + ((60 * sx * sy) >> 28)  # arithmetic shift == floor
        cx = w // 2 + 3 * frame
        cy = h // 2 + 2 * frame
        rad = h // 5
        base = base + 40 * (((x - cx) ** 2 + (y - cy) ** 2) < rad * rad)
        for ch in range(3):
            s = np.zeros((h, w), dtype=np.int64)
            for k in range(4):
                s += ((hs >> np.uint64(20 * ch + 5 * k)) & np.uint64(31)).astype(np.int64)
            n = ((s * 55 + 128) >> 8) - 13
            out[..., ch] = np.clip(base + n + 10 * (ch - 1), 0, 255).astype(np.uint8)
        return out
    if kid == 2:  # ramp
        g = ((x + 2 * y + frame) & 255).astype(np.uint8)
        out[...] = g[..., None]
        return out
    if kid == 3:  # flat
        out[..., 0] = 77
        out[..., 1] = 150
        out[..., 2] = 201
        return out
    if kid == 4:  # tie
        g = (100 + (hs & np.uint64(1)).astype(np.int64)).astype(np.uint8)
        out[...] = g[..., None]
        return out
    # step
    g = 60 + 90 * ((((x + frame) >> 4) + (y >> 4)) & 1) + ((hs & np.uint64(3)).astype(np.int64) - 1)
    out[...] = np.clip(g, 0, 255).astype(np.uint8)[..., None]
    return out


def synth_clip(kind: str, seed: int, video: int, n_frames: int, h: int, w: int, frame0: int = 0) -> np.ndarray:
    """[n_frames][h][w][3] uint8 BGR."""
    return np.stack([synth_frame(kind, seed, video, frame0 + f, h, w) for f in range(n_frames)])
