"""TEST INFRASTRUCTURE (oracle) -- camera-motion estimation, the "warp" in the reference's warped optical flow.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file; the product never does.

What the reference does (src/features_GPU_compute/build_wof_clips.py:55-76): it shells out to dense_flow's
``extract_warp_gpu`` (third-party, not in /root/reference, no version pinned).  That tool follows improved dense
trajectories (Wang & Schmid, ICCV 2013, section 3.1 and its public code): per frame pair it (1) computes TV-L1 flow,
(2) collects matches prev -> current from two sources -- SURF descriptors, and ``goodFeaturesToTrack`` corners of the
previous frame moved by the flow (``MatchFromFlow``: 1000 corners, quality 0.001, min distance 3) -- (3) fits a homography
with ``findHomography(prev_pts, pts, RANSAC, 1)`` and keeps it only when there were more than 50 matches and more than 25
inliers, (4) warps the current frame by the inverse homography and (5) computes the flow again.

PARITY UNPINNED: no frames, flow images or binary exist in the reference.  This file restates steps (2)-(4) for the
FLOW-MATCH source only (SURF is not restated) from the published descriptions of the OpenCV functions involved:

* ``cv::cornerMinEigenVal`` (block 3, Sobel aperture 3, BORDER_REFLECT_101): derivatives scaled by 1 / (2^(3-1) * 3 * 255),
  products summed over the 3x3 block, ``(a/2 + c/2) - sqrt((a/2 - c/2)^2 + b^2)``;
* ``cv::goodFeaturesToTrack``: zero everything <= quality * max, keep interior pixels equal to their 3x3 maximum, sort by
  strength (descending; equal strengths: later pixel first), greedily drop a corner closer than minDistance to a kept one;
* ``cv::findHomography(RANSAC)``: minimal 4-point samples that keep the orientation of every point triple, forward
  reprojection error against threshold^2, most inliers wins, re-estimation on the inliers (normalised least squares; cv's
  final Levenberg-Marquardt polish is not restated).  cv draws its samples from its own RNG; here the draw is a counter
  hash so that device and oracle can take the SAME samples -- the set of hypotheses is a free choice of any RANSAC.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import numpy as np

F = np.float32
MAX_CORNERS, QUALITY, MIN_DISTANCE = 1000, 0.001, 3.0       # dense_flow / IDT MatchFromFlow
RANSAC_THRESHOLD, MIN_MATCHES, MIN_INLIERS = 1.0, 50, 25    # findHomography(..., RANSAC, 1); "> 50" and "> 25" guards


def _r101(i: int, n: int) -> int:
    return -i if i < 0 else (2 * n - 2 - i if i >= n else i)


def corner_strength(img: np.ndarray) -> np.ndarray:
    """Smaller eigenvalue of the 3x3-block gradient covariance, float32, sums in row-then-column order."""
    h, w = img.shape
    ry = np.array([_r101(i, h) for i in range(-2, h + 2)])
    rx = np.array([_r101(i, w) for i in range(-2, w + 2)])
    # reflect twice, exactly as the filters do: the block reads derivative (yy, xx) = reflect(y + dy), the derivative reads
    # source pixels reflect(yy +- 1)
    src = img.astype(np.int32)

    def deriv(yy: np.ndarray, xx: np.ndarray):
        y0 = np.array([_r101(int(v) - 1, h) for v in yy])[:, None]
        y2 = np.array([_r101(int(v) + 1, h) for v in yy])[:, None]
        x0 = np.array([_r101(int(v) - 1, w) for v in xx])[None, :]
        x2 = np.array([_r101(int(v) + 1, w) for v in xx])[None, :]
        y1, x1 = yy[:, None], xx[None, :]
        gx = (src[y0, x2] + 2 * src[y1, x2] + src[y2, x2]) - (src[y0, x0] + 2 * src[y1, x0] + src[y2, x0])
        gy = (src[y2, x0] + 2 * src[y2, x1] + src[y2, x2]) - (src[y0, x0] + 2 * src[y0, x1] + src[y0, x2])
        scale = F(1.0 / (4.0 * 3.0 * 255.0))
        return gx.astype(F) * scale, gy.astype(F) * scale

    a = np.zeros((h, w), F)
    b = np.zeros((h, w), F)
    c = np.zeros((h, w), F)
    for dy in (-1, 0, 1):
        ra = np.zeros((h, w), F)
        rb = np.zeros((h, w), F)
        rc = np.zeros((h, w), F)
        yy = ry[2 + dy:2 + dy + h]
        for dx in (-1, 0, 1):
            xx = rx[2 + dx:2 + dx + w]
            fx, fy = deriv(yy, xx)
            ra = ra + fx * fx
            rb = rb + fx * fy
            rc = rc + fy * fy
        a, b, c = a + ra, b + rb, c + rc
    a = a * F(0.5)
    c = c * F(0.5)
    d = a - c
    return ((a + c) - np.sqrt(d * d + b * b, dtype=F)).astype(F)


def corner_peaks(strength: np.ndarray) -> np.ndarray:
    """Interior pixels that equal their 3x3 maximum keep their strength, everything else is 0."""
    h, w = strength.shape
    out = np.zeros_like(strength)
    for y in range(1, h - 1):
        for x in range(1, w - 1):
            v = strength[y, x]
            if v == strength[y - 1:y + 2, x - 1:x + 2].max():
                out[y, x] = v
    return out


def good_features(img: np.ndarray, max_corners: int = MAX_CORNERS, quality: float = QUALITY, min_distance: float = MIN_DISTANCE) -> np.ndarray:
    """cv::goodFeaturesToTrack on an 8-bit frame -> [k, 2] float32 (x, y), strongest first."""
    s = corner_strength(img)
    top = max(float(s.max()), 0.0)
    peaks = corner_peaks(s)
    h, w = s.shape
    thresh = F(F(top) * F(quality))
    cand = [i for i in range(h * w) if peaks.flat[i] > thresh and peaks.flat[i] != 0]
    cand.sort(key=lambda i: (-float(peaks.flat[i]), -i))
    kept: List[Tuple[int, int]] = []
    md2 = F(min_distance) * F(min_distance)
    for i in cand:
        y, x = divmod(i, w)
        if min_distance >= 1 and any(F(x - kx) * F(x - kx) + F(y - ky) * F(y - ky) < md2 for kx, ky in kept):
            continue
        kept.append((x, y))
        if len(kept) == max_corners:
            break
    return np.array(kept, dtype=F).reshape(-1, 2)


def matches_from_flow(corners: np.ndarray, u1: np.ndarray, u2: np.ndarray) -> np.ndarray:
    """MatchFromFlow: every corner (integer pixel) moves by the flow at that pixel."""
    h, w = u1.shape
    out = np.empty_like(corners, dtype=F)
    for k, (x, y) in enumerate(corners):
        xi = min(max(int(round(float(x))), 0), w - 1)
        yi = min(max(int(round(float(y))), 0), h - 1)
        out[k] = (F(xi) + u1[yi, xi], F(yi) + u2[yi, xi])
    return out


def mix32(x: int) -> int:
    x &= 0xFFFFFFFF
    x ^= x >> 16
    x = (x * 0x7FEB352D) & 0xFFFFFFFF
    x ^= x >> 15
    x = (x * 0x846CA68B) & 0xFFFFFFFF
    x ^= x >> 16
    return x


def draw_sample(seed: int, pair: int, hypothesis: int, n: int) -> Optional[List[int]]:
    """Four distinct match indices of hypothesis `hypothesis` of match set `pair`."""
    base = mix32(mix32(seed + pair) + hypothesis)
    idx: List[int] = []
    for k in range(4):
        for a in range(64):
            cand = mix32(base + 16 * a + k) % n
            if cand not in idx:
                idx.append(cand)
                break
        else:
            return None
    return idx


def _orient(a, b, c) -> float:
    return (float(b[0]) - float(a[0])) * (float(c[1]) - float(a[1])) - (float(b[1]) - float(a[1])) * (float(c[0]) - float(a[0]))


def homography_4pt(s: np.ndarray, d: np.ndarray) -> Optional[np.ndarray]:
    A = np.zeros((8, 8))
    r = np.zeros(8)
    for k in range(4):
        x, y, u, v = float(s[k, 0]), float(s[k, 1]), float(d[k, 0]), float(d[k, 1])
        A[2 * k] = [x, y, 1, 0, 0, 0, -u * x, -u * y]
        A[2 * k + 1] = [0, 0, 0, x, y, 1, -v * x, -v * y]
        r[2 * k], r[2 * k + 1] = u, v
    try:
        hv = np.linalg.solve(A, r)
    except np.linalg.LinAlgError:
        return None
    return np.append(hv, 1.0).reshape(3, 3)


def reprojection_error2(H: np.ndarray, src: np.ndarray, dst: np.ndarray) -> np.ndarray:
    x, y = src[:, 0].astype(np.float64), src[:, 1].astype(np.float64)
    wq = H[2, 0] * x + H[2, 1] * y + 1.0
    ex = (H[0, 0] * x + H[0, 1] * y + H[0, 2]) / wq - dst[:, 0].astype(np.float64)
    ey = (H[1, 0] * x + H[1, 1] * y + H[1, 2]) / wq - dst[:, 1].astype(np.float64)
    return ex * ex + ey * ey


def refit_homography(src: np.ndarray, dst: np.ndarray) -> np.ndarray:
    """Normalised least squares with h33 = 1 in the normalised frame (Hartley): solved here by an orthogonal method."""
    s, d = src.astype(np.float64), dst.astype(np.float64)
    cs, cd = s.mean(0), d.mean(0)
    ss = np.sqrt(2.0) * len(s) / np.sqrt(((s - cs) ** 2).sum(1)).sum()
    sd = np.sqrt(2.0) * len(d) / np.sqrt(((d - cd) ** 2).sum(1)).sum()
    sn, dn = (s - cs) * ss, (d - cd) * sd
    rows, rhs = [], []
    for (x, y), (u, v) in zip(sn, dn):
        rows.append([x, y, 1, 0, 0, 0, -u * x, -u * y])
        rows.append([0, 0, 0, x, y, 1, -v * x, -v * y])
        rhs += [u, v]
    hv = np.linalg.lstsq(np.array(rows), np.array(rhs), rcond=None)[0]
    Hn = np.append(hv, 1.0).reshape(3, 3)
    Ts = np.array([[ss, 0, -ss * cs[0]], [0, ss, -ss * cs[1]], [0, 0, 1]])
    Td = np.array([[sd, 0, -sd * cd[0]], [0, sd, -sd * cd[1]], [0, 0, 1]])
    H = np.linalg.inv(Td) @ Hn @ Ts
    return H / H[2, 2]


def ransac_homography(src: np.ndarray, dst: np.ndarray, threshold: float = RANSAC_THRESHOLD, hypotheses: int = 512, seed: int = 0, pair: int = 0,
                      refit: bool = True):
    """-> (H [3,3], inliers, winner index, mask [n] uint8).  Identity / 0 / -1 when nothing valid was drawn."""
    n = len(src)
    best = (-1, -1, np.eye(3))
    thr2 = float(F(threshold)) ** 2
    if n >= 4:
        for j in range(hypotheses):
            idx = draw_sample(seed, pair, j, n)
            if idx is None:
                continue
            s4, d4 = src[idx], dst[idx]
            if not all(_orient(s4[a], s4[(a + 1) & 3], s4[(a + 2) & 3]) * _orient(d4[a], d4[(a + 1) & 3], d4[(a + 2) & 3]) > 0 for a in range(4)):
                continue
            H = homography_4pt(s4, d4)
            if H is None:
                continue
            cnt = int((reprojection_error2(H, src, dst) <= thr2).sum())
            if cnt > best[0]:
                best = (cnt, j, H)
    cnt, j, H = best
    if cnt < 0:
        return np.eye(3), 0, -1, np.zeros(n, np.uint8)
    mask = (reprojection_error2(H, src, dst) <= thr2).astype(np.uint8)
    if refit and cnt >= 4:
        H = refit_homography(src[mask == 1], dst[mask == 1])
    return H, cnt, j, mask


def camera_motion(prev: np.ndarray, u1: np.ndarray, u2: np.ndarray, seed: int = 0, pair: int = 0, hypotheses: int = 512) -> Tuple[np.ndarray, int, int]:
    """Steps (2)-(3) for one pair: -> (H prev->current, or identity when the guards fail; matches; inliers)."""
    corners = good_features(prev)
    if len(corners) <= MIN_MATCHES:
        return np.eye(3), len(corners), 0
    H, inliers, _, _ = ransac_homography(corners, matches_from_flow(corners, u1, u2), RANSAC_THRESHOLD, hypotheses, seed, pair)
    return (H if inliers > MIN_INLIERS else np.eye(3)), len(corners), inliers
