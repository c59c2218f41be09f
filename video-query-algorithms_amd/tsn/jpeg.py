"""Baseline JPEG decoding of a batch of equally sized frames: Huffman decoding on host threads inside the library,
IDCT / chroma upsampling / colour transform on the GPU (csrc/vq_jpeg.hip).

Replaces ``cv2.imread(..., IMREAD_COLOR)`` / ``cv2.imread(..., IMREAD_GRAYSCALE)`` of the reference's frame loops
(src/features_GPU_compute/calcSig_wOF.py:92,105-106) for the files its frame preparation writes; the pixels are the ones
libjpeg(-turbo) produces (oracle/jpeg_oracle.py, pinned bit for bit against Pillow's libjpeg-turbo).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Sequence, Tuple, Union

import numpy as np

from .. import _lib
from .._lib import call


def info(data: Union[bytes, str]) -> Tuple[int, int, int]:
    """(height, width, components) of a JPEG file's frame header (file contents, or a path)."""
    h, w, c = C.c_int32(), C.c_int32(), C.c_int32()
    if isinstance(data, str):
        call("vq_jpeg_info_file", os.fsencode(data), C.byref(h), C.byref(w), C.byref(c))
    else:
        call("vq_jpeg_info", data, len(data), C.byref(h), C.byref(w), C.byref(c))
    return h.value, w.value, c.value


class JpegDecoder:
    def __init__(self, max_frames: int, max_h: int, max_w: int, device: int = 0):
        self.max_frames, self.max_h, self.max_w, self.device = int(max_frames), int(max_h), int(max_w), int(device)
        self._h = C.c_void_p()
        call("vq_jpeg_create", self.max_frames, self.max_h, self.max_w, self.device, C.byref(self._h))

    def _submit(self, files: Sequence[Union[bytes, str]], color: bool, out_host, want_dev: bool, stream: int = 0, planes_only: bool = False):
        if len(files) and all(isinstance(f, str) for f in files):      # paths only: the library's threads read the files
            n = len(files)
            h, w, _ = info(files[0])
            joined = "\0".join(files)                                  # one object, not n pointers (3-4 ms per 8 000 under the interpreter lock)
            if joined.count("\0") != n - 1:
                raise ValueError("a path contains a NUL character")
            paths = os.fsencode(joined + "\0")
            dev = C.c_void_p()
            host = np.empty((n, h, w, 3) if color else (n, h, w), dtype=np.uint8) if out_host else None
            call("vq_jpeg_decode_path_list", self._h, paths, len(paths), n, int(bool(color)) | (2 if planes_only else 0), h, w, host.ctypes.data_as(C.c_void_p) if host is not None else None,
                 C.byref(dev) if want_dev else None, C.c_void_p(stream) if stream else None)
            return host, dev.value, (n, h, w)
        blobs: List[bytes] = []
        for f in files:
            if isinstance(f, (bytes, bytearray, memoryview)):
                blobs.append(bytes(f))
            else:
                with open(f, "rb") as fh:
                    blobs.append(fh.read())
        n = len(blobs)
        if n == 0:
            raise ValueError("no files")
        h, w, _ = info(blobs[0])
        ptrs = (C.c_char_p * n)(*blobs)
        sizes = (C.c_int64 * n)(*[len(b) for b in blobs])
        dev = C.c_void_p()
        host = None
        if out_host:
            host = np.empty((n, h, w, 3) if color else (n, h, w), dtype=np.uint8)
        call("vq_jpeg_decode", self._h, ptrs, sizes, n, int(bool(color)) | (2 if planes_only else 0), h, w, host.ctypes.data_as(C.c_void_p) if host is not None else None,
             C.byref(dev) if want_dev else None, C.c_void_p(stream) if stream else None)
        return host, dev.value, (n, h, w)

    def decode(self, files: Sequence[Union[bytes, str]], color: bool = True) -> np.ndarray:
        """File contents or paths -> uint8 [n, h, w, 3] (B, G, R: cv2 order) or [n, h, w] (grey: the Y plane)."""
        return self._submit(files, color, True, False)[0]

    def decode_to_device(self, files: Sequence[Union[bytes, str]], color: bool = True, stream: int = 0):
        """-> (device pointer of [n, h, w, 3 | 1] uint8, (n, h, w)); the memory belongs to the decoder and is valid until
        its next call -- feed it to ``vq_resize_crop(frames_on_device=1)``.  ``stream``: the HIP stream the copies and kernels run
        on (the call returns when they are done)."""
        _, dev, shape = self._submit(files, color, False, True, stream)
        return dev, shape

    def decode_to_crops(self, files: Sequence[Union[bytes, str]], color: bool, crop: int, crops_dev_ptr: int, planes: int = 10, stream: int = 0):
        """Frames that already have the resize size: decode and write the top-left ``crop`` x ``crop`` pixels -- what resize + crop 0 leaves
        of them -- into device memory without the whole-frame pixel pass (vq_jpeg_crops).  Colour: [n, crop, crop, 3] B, G, R; grey:
        the files plane-major (all frames of plane 0, then plane 1, ...), [n / planes, crop, crop, planes].  Returns (n, h, w)."""
        _, _, shape = self._submit(files, color, False, False, stream, planes_only=True)
        call("vq_jpeg_crops", self._h, 3 if color else int(planes), int(crop), C.c_void_p(crops_dev_ptr), C.c_void_p(stream) if stream else None)
        return shape

    def close(self):
        if self._h:
            _lib.load().vq_jpeg_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
