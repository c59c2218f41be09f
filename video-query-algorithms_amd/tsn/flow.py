"""TV-L1 optical flow of frame pairs on the GPU: the flow half of the reference's frame preparation.

The reference gets its ``flow_x_NNNNN.jpg`` / ``flow_y_NNNNN.jpg`` frames from a third-party binary
(``extract_warp_gpu -b 20 -t 1 -s 1``, src/features_GPU_compute/build_wof_clips.py:55-76) and then regroups them into
clips (:78-128, ``frames.clip_plan``).  ``Tvl1Flow`` is the handle over ``vq_flow_*`` (csrc/vq_flow.hip): batches of
grey frame pairs in, fp32 flow fields and / or the 8-bit flow images out.  Parity is unpinned (no binary, no frames, no
flow images in the reference); the kernels follow oracle/tvl1_oracle.py, the published algorithm with OpenCV's defaults.
The SURF + RANSAC homography of dense_flow's "warp" step is not built -- pass ``homographies`` to compensate a known
camera motion.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np

from .. import _lib
from .._lib import Tvl1Params, call


class Tvl1Flow:
    def __init__(self, max_pairs: int, h: int, w: int, device: int = 0, **params):
        """params: tau, lambda_, theta, epsilon, scale_step, nscales, warps, iterations, bound (defaults: OpenCV's)."""
        p = Tvl1Params()
        call("vq_tvl1_default_params", C.byref(p))
        for k, v in params.items():
            if not hasattr(p, k):
                raise TypeError("unknown TV-L1 parameter %r" % k)
            setattr(p, k, v)
        self.params = p
        self.max_pairs, self.h, self.w, self.device = int(max_pairs), int(h), int(w), int(device)
        self._h = C.c_void_p()
        call("vq_flow_create", self.max_pairs, self.h, self.w, C.byref(p), self.device, C.byref(self._h))
        n = C.c_int32()
        sizes = np.zeros((16, 2), dtype=np.int32)
        call("vq_flow_levels", self._h, C.byref(n), sizes.ctypes.data_as(C.POINTER(C.c_int32)), 16)
        self.levels = [tuple(int(v) for v in sizes[i]) for i in range(n.value)]          # (h, w), finest first

    def flow(self, frames0: np.ndarray, frames1: np.ndarray, homographies: Optional[np.ndarray] = None, images: bool = True,
             fields: bool = True, iterations: bool = False):
        """frames0 / frames1: uint8 [n, h, w].  Returns a dict with ``u1``, ``u2`` (fp32 dx, dy), ``flow_x``, ``flow_y``
        (uint8 images as extract_warp_gpu writes them) and ``iters`` [levels, warps, n] as requested."""
        f0 = np.ascontiguousarray(frames0, dtype=np.uint8)
        f1 = np.ascontiguousarray(frames1, dtype=np.uint8)
        if f0.shape != f1.shape or f0.ndim != 3 or f0.shape[1:] != (self.h, self.w):
            raise ValueError("frames must both be [n, %d, %d] uint8" % (self.h, self.w))
        n = f0.shape[0]
        hm = None
        if homographies is not None:
            hm = np.ascontiguousarray(homographies, dtype=np.float64).reshape(n, 9)
        out = {}
        if fields:
            out["u1"] = np.empty((n, self.h, self.w), dtype=np.float32)
            out["u2"] = np.empty((n, self.h, self.w), dtype=np.float32)
        if images:
            out["flow_x"] = np.empty((n, self.h, self.w), dtype=np.uint8)
            out["flow_y"] = np.empty((n, self.h, self.w), dtype=np.uint8)
        if iterations:
            out["iters"] = np.zeros((len(self.levels), self.params.warps, n), dtype=np.int32)

        def ptr(key):
            return out[key].ctypes.data_as(C.c_void_p) if key in out else None
        call("vq_flow_tvl1", self._h, f0.ctypes.data_as(C.c_void_p), f1.ctypes.data_as(C.c_void_p), 0, n,
             hm.ctypes.data_as(C.c_void_p) if hm is not None else None, ptr("u1"), ptr("u2"), ptr("flow_x"), ptr("flow_y"), ptr("iters"), None)
        return out

    def consecutive(self, frames: np.ndarray):
        """Grey frames [n + 1, h, w] of a video -> (flow_x, flow_y) uint8 [n, h, w] between consecutive frames
        (``extract_warp_gpu -s 1``), in batches of ``max_pairs``."""
        fx, fy = [], []
        for i in range(0, frames.shape[0] - 1, self.max_pairs):
            j = min(i + self.max_pairs, frames.shape[0] - 1)
            r = self.flow(frames[i:j], frames[i + 1:j + 1], fields=False)
            fx.append(r["flow_x"])
            fy.append(r["flow_y"])
        return np.concatenate(fx), np.concatenate(fy)

    def close(self):
        if self._h:
            _lib.load().vq_flow_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
