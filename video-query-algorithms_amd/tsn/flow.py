"""TV-L1 optical flow of frame pairs on the GPU: the flow half of the reference's frame preparation.

The reference gets its ``flow_x_NNNNN.jpg`` / ``flow_y_NNNNN.jpg`` frames from a third-party binary
(``extract_warp_gpu -b 20 -t 1 -s 1``, src/features_GPU_compute/build_wof_clips.py:55-76) and then regroups them into
clips (:78-128, ``frames.clip_plan``).  ``Tvl1Flow`` is the handle over ``vq_flow_*`` (csrc/vq_flow.hip): batches of
grey frame pairs in, fp32 flow fields and / or the 8-bit flow images out.  Parity is unpinned (no binary, no frames, no
flow images in the reference); the kernels follow oracle/tvl1_oracle.py, the published algorithm with OpenCV's defaults.
``warped`` adds the camera-motion compensation of dense_flow's "warp" step in its flow-match form (improved dense
trajectories): Shi-Tomasi corners of the first frame moved by the first-pass flow, a RANSAC homography over those
matches, the second frame warped back by it, the flow computed again.  The SURF matches the binary merges into the same
RANSAC are not built.
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

    def last_timing(self):
        """(device ms of the inner loops, iteration-kernel launches) of the last ``flow`` call (HIP events inside the library)."""
        ms, n = C.c_double(), C.c_int32()
        call("vq_flow_last_timing", self._h, C.byref(ms), C.byref(n))
        return ms.value, n.value

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

    # ---- camera-motion compensation ("warp"): corners -> flow matches -> RANSAC homography -> second pass
    MAX_CORNERS, QUALITY, MIN_DISTANCE = 1000, 0.001, 3.0     # dense_flow's MatchFromFlow
    RANSAC_THRESHOLD, MIN_MATCHES, MIN_INLIERS = 1.0, 50, 25  # findHomography(..., RANSAC, 1) and its two guards

    def good_features(self, frames: np.ndarray, max_corners: int = MAX_CORNERS, quality: float = QUALITY,
                      min_distance: float = MIN_DISTANCE):
        """uint8 [n, h, w] -> (corners float32 [n, max_corners, 2] as (x, y), counts int32 [n]); strongest first."""
        f = np.ascontiguousarray(frames, dtype=np.uint8)
        if f.ndim != 3 or f.shape[1:] != (self.h, self.w):
            raise ValueError("frames must be [n, %d, %d] uint8" % (self.h, self.w))
        n = f.shape[0]
        corners = np.zeros((n, int(max_corners), 2), dtype=np.float32)
        counts = np.zeros(n, dtype=np.int32)
        call("vq_flow_good_features", self._h, f.ctypes.data_as(C.c_void_p), 0, n, int(max_corners), float(quality), float(min_distance),
             corners.ctypes.data_as(C.c_void_p), counts.ctypes.data_as(C.c_void_p), None)
        return corners, counts

    def ransac_homography(self, src: np.ndarray, dst: np.ndarray, counts: np.ndarray, threshold: float = RANSAC_THRESHOLD,
                          hypotheses: int = 512, seed: int = 0, refit: bool = True):
        """Match sets src -> dst (float32 [n, max_points, 2], counts [n]) -> dict(H [n,3,3], inliers [n], winner [n],
        mask [n, max_points])."""
        a = np.ascontiguousarray(src, dtype=np.float32)
        b = np.ascontiguousarray(dst, dtype=np.float32)
        c = np.ascontiguousarray(counts, dtype=np.int32)
        if a.shape != b.shape or a.ndim != 3 or a.shape[2] != 2 or c.shape != (a.shape[0],):
            raise ValueError("src / dst must be [n, max_points, 2] and counts [n]")
        n, mp = a.shape[0], a.shape[1]
        out = {"H": np.zeros((n, 3, 3)), "inliers": np.zeros(n, np.int32), "winner": np.zeros(n, np.int32), "mask": np.zeros((n, mp), np.uint8)}
        call("vq_flow_ransac_homography", self._h, a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p), c.ctypes.data_as(C.c_void_p),
             n, mp, float(threshold), int(hypotheses), int(seed) & 0xFFFFFFFF, int(bool(refit)), out["H"].ctypes.data_as(C.c_void_p),
             out["inliers"].ctypes.data_as(C.c_void_p), out["winner"].ctypes.data_as(C.c_void_p), out["mask"].ctypes.data_as(C.c_void_p), None)
        return out

    def camera_motion(self, frames0: np.ndarray, u1: np.ndarray, u2: np.ndarray, seed: int = 0, hypotheses: int = 512):
        """Homography frames0 -> frames1 per pair from corners of frames0 moved by the flow (u1, u2); identity where the
        guards of dense_flow fail (<= 50 matches or <= 25 inliers).  -> (H [n,3,3], matches [n], inliers [n])."""
        corners, counts = self.good_features(frames0)
        n = corners.shape[0]
        xi = np.clip(np.rint(corners[..., 0]).astype(np.int64), 0, self.w - 1)
        yi = np.clip(np.rint(corners[..., 1]).astype(np.int64), 0, self.h - 1)
        pair = np.arange(n)[:, None]
        moved = np.stack([xi.astype(np.float32) + u1[pair, yi, xi], yi.astype(np.float32) + u2[pair, yi, xi]], axis=-1)
        r = self.ransac_homography(corners, moved, counts, seed=seed, hypotheses=hypotheses)
        H = r["H"].copy()
        H[(counts <= self.MIN_MATCHES) | (r["inliers"] <= self.MIN_INLIERS)] = np.eye(3)
        return H, counts, r["inliers"]

    def warped(self, frames0: np.ndarray, frames1: np.ndarray, seed: int = 0, images: bool = True, fields: bool = False):
        """The warped flow of extract_warp_gpu (flow-match branch): first-pass flow -> camera homography -> frames1 warped
        back by it -> flow again, in ONE library call (``vq_flow_warped``: the frames go up once, the first-pass fields never
        leave the device).  Returns flow()'s dict plus ``H`` (frames0 -> frames1), ``matches`` and ``inliers``."""
        f0 = np.ascontiguousarray(frames0, dtype=np.uint8)
        f1 = np.ascontiguousarray(frames1, dtype=np.uint8)
        if f0.shape != f1.shape or f0.ndim != 3 or f0.shape[1:] != (self.h, self.w):
            raise ValueError("frames must both be [n, %d, %d] uint8" % (self.h, self.w))
        n = f0.shape[0]
        out = {"H": np.zeros((n, 3, 3)), "matches": np.zeros(n, np.int32), "inliers": np.zeros(n, np.int32)}
        if fields:
            out["u1"] = np.empty((n, self.h, self.w), dtype=np.float32)
            out["u2"] = np.empty((n, self.h, self.w), dtype=np.float32)
        if images:
            out["flow_x"] = np.empty((n, self.h, self.w), dtype=np.uint8)
            out["flow_y"] = np.empty((n, self.h, self.w), dtype=np.uint8)

        def ptr(key):
            return out[key].ctypes.data_as(C.c_void_p) if key in out else None
        call("vq_flow_warped", self._h, f0.ctypes.data_as(C.c_void_p), f1.ctypes.data_as(C.c_void_p), n, int(seed) & 0xFFFFFFFF, 512, ptr("u1"), ptr("u2"),
             ptr("flow_x"), ptr("flow_y"), ptr("H"), ptr("matches"), ptr("inliers"), None)
        return out

    def warped_steps(self, frames0: np.ndarray, frames1: np.ndarray, seed: int = 0, images: bool = True, fields: bool = False):
        """The same sequence call by call through the public pieces (flow, good_features, ransac_homography, flow): what
        ``warped`` is tested against."""
        first = self.flow(frames0, frames1, images=False, fields=True)
        H, matches, inliers = self.camera_motion(frames0, first["u1"], first["u2"], seed=seed)
        # flow(homographies=G) shows the second frame as out(x) = frame1(G^-1 x); the compensated frame is frame1(H x)
        out = self.flow(frames0, frames1, homographies=np.linalg.inv(H), images=images, fields=fields)
        out.update(H=H, matches=matches, inliers=inliers)
        return out

    def warped_consecutive(self, frames: np.ndarray, seed: int = 0):
        """Grey frames [n + 1, h, w] -> (flow_x, flow_y) uint8 [n, h, w]: the files extract_warp_gpu -s 1 writes."""
        fx, fy = [], []
        for i in range(0, frames.shape[0] - 1, self.max_pairs):
            j = min(i + self.max_pairs, frames.shape[0] - 1)
            r = self.warped(frames[i:j], frames[i + 1:j + 1], seed=seed + i)
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
