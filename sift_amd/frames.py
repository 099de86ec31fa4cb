"""The step before the texture path: frames in the layout the reference's frame source produces.

``extract_frames`` (/root/reference/app/video_processor.py:18-49) hands the checkers a Python list of
``[H][W][3]`` uint8 BGR, C-contiguous arrays, at most ``max_frames`` of them (the service asks for 450,
/root/reference/app/main.py:393).  Decoding video is not part of this path; what matters to it is where
those bytes live, because a host-fed GPU pass runs at the speed of the host-to-device copy:
pinned memory moves at PCIe speed, pageable memory at about half of it.

``PinnedFrameBatch`` is one page-locked ``[N][H][W][3]`` buffer owned by the library
(``stx_pinned_alloc``), exposed as a numpy array; ``frames()`` gives the list-of-arrays view the
reference's callers expect (views, no copies).  ``read_raw_bgr`` fills it from a raw BGR24 stream
(what ``ffmpeg -f rawvideo -pix_fmt bgr24`` or any capture loop writes) without an intermediate copy,
and ``analyze`` runs the batch through ``stx_analyze_batch`` straight from the pinned pages.
"""
from __future__ import annotations

import ctypes as C
from typing import BinaryIO, List, Optional

import numpy as np

from ._lib import StxParams
from .texture import TextureContext

MAX_SERVICE_FRAMES = 450  # main.py:393


class _PinnedPages:
    """Owner of one page-locked allocation, exported through the buffer protocol: numpy arrays built on it keep
    it (and through it the context) alive, and the memory goes back to the library when the last one dies."""

    def __init__(self, ctx: TextureContext, nbytes: int):
        self.ctx = ctx
        p = C.c_void_p()
        ctx._check(ctx.lib.stx_pinned_alloc(ctx.handle, nbytes, C.byref(p)), "stx_pinned_alloc")
        self._ptr = p
        self._buf = (C.c_uint8 * nbytes).from_address(p.value)

    @property
    def __array_interface__(self):
        return {"shape": (len(self._buf),), "typestr": "|u1", "data": (self._ptr.value, False), "version": 3}

    def __del__(self):
        try:
            if self._ptr and self.ctx.handle:
                self.ctx.lib.stx_pinned_free(self.ctx.handle, self._ptr)
            self._ptr = None
        except Exception:
            pass


class PinnedFrameBatch:
    """A reusable page-locked frame buffer in the ``extract_frames`` layout."""

    def __init__(self, ctx: TextureContext, capacity: int, height: int, width: int):
        if capacity < 1 or height < 1 or width < 1:
            raise ValueError("capacity, height and width must be positive")
        self.ctx = ctx
        self.capacity, self.height, self.width = int(capacity), int(height), int(width)
        self.nbytes = self.capacity * self.height * self.width * 3
        self._pages = _PinnedPages(ctx, self.nbytes)
        # every numpy view handed out (array, frames()) has the pages object as its base, so the page-locked
        # memory is released only after the batch AND the last such view are gone
        self.array = np.asarray(self._pages).reshape(self.capacity, self.height, self.width, 3)
        self.count = 0  # frames currently held

    def close(self):
        """Drop the batch's own references; the pages are freed when no view of them is left."""
        self.array = None
        self._pages = None

    # ---- filling
    def clear(self):
        self.count = 0

    def append(self, frame: np.ndarray) -> bool:
        """Copy one BGR frame in; False when the batch is full."""
        if self.count >= self.capacity:
            return False
        a = np.asarray(frame)
        if a.shape != (self.height, self.width, 3) or a.dtype != np.uint8:
            raise ValueError(f"expected a {self.height} x {self.width} x 3 uint8 frame, got {a.shape} {a.dtype}")
        self.array[self.count] = a
        self.count += 1
        return True

    def read_raw_bgr(self, stream: BinaryIO, max_frames: Optional[int] = None) -> int:
        """Fill the batch from a raw BGR24 stream (frame after frame, rows top to bottom, no padding), reading
        straight into the pinned pages.  Stops at end of stream, at ``max_frames`` (the reference's
        ``max_frames`` semantics, video_processor.py:44-45) or when the batch is full; a trailing partial
        frame is dropped, as a decoder drops a truncated last frame.  Returns the number of frames now held."""
        limit = self.capacity if not max_frames else min(self.capacity, int(max_frames))
        frame_bytes = self.height * self.width * 3
        flat = self.array.reshape(self.capacity, frame_bytes)
        while self.count < limit:
            view = memoryview(flat[self.count])
            got = 0
            while got < frame_bytes:
                n = stream.readinto(view[got:])
                if not n:
                    break
                got += n
            if got < frame_bytes:
                break
            self.count += 1
        return self.count

    # ---- consuming
    def frames(self) -> List[np.ndarray]:
        """The list the reference's checkers take: one [H][W][3] view per held frame."""
        return [self.array[i] for i in range(self.count)]

    def analyze(self, params: Optional[StxParams] = None):
        """Per-frame records for the held frames (host-fed pass: copy, gray, all five analyses)."""
        return self.ctx.analyze_batch(self.array[:self.count], params)
