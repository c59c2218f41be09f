"""TEST INFRASTRUCTURE (oracle) -- TV-L1 optical flow, the arithmetic behind the reference's flow frames.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file; the product never does.

What the reference does (src/features_GPU_compute/build_wof_clips.py:55-76): per video it shells out to
``<TSN_ROOT>/lib/dense_flow/build/extract_warp_gpu -f <video> -x flow_x -y flow_y -b 20 -t 1 -d <gpu> -s 1 -o dir``
and later regroups the ``flow_x_NNNNN.jpg`` / ``flow_y_NNNNN.jpg`` files into clips (:78-128).  The binary is third-party
(yjxiong/dense_flow on OpenCV's CUDA module; no commit pinned, not in /root/reference, cannot be built or run here):
``-t 1`` = TV-L1 (cv::cuda::OpticalFlowDual_TVL1 with its default parameters), ``-b 20`` = flow values are clamped to
[-20, 20] pixels and quantised to 8 bits, ``-s 1`` = consecutive frames.  "Warp" = the camera-motion compensation of
improved dense trajectories (Wang & Schmid 2013): a homography is fitted (RANSAC) to SURF matches plus flow matches, the
second frame is warped by it and the flow is computed again.

PARITY UNPINNED.  Nothing the reference holds pins any of this: no frames, no flow images, no binary.  This file
restates the PUBLISHED algorithm -- Zach, Pock & Bischof, "A duality based approach for realtime TV-L1 optical flow"
(DAGM 2007) in the formulation of Sanchez Perez, Meinhardt-Llopis & Facciolo, "TV-L1 Optical Flow Estimation" (IPOL 2013,
Algorithm 1 and its reference code), which OpenCV's implementation follows -- with OpenCV's published default parameters
and its interpolation choices (bilinear warping, bilinear pyramid).  The estimation of the warp step's matrix is restated in
oracle/warp_oracle.py (flow-match branch; SURF is not restated): ``warp_homography`` here takes the 3x3 matrix as an input.

Everything is float32 with the operation order written out, so that a device kernel can follow it operation for operation.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np

# cv::cuda::OpticalFlowDual_TVL1::create() defaults (OpenCV documentation)
TAU, LAMBDA, THETA = 0.25, 0.15, 0.3
NSCALES, WARPS, EPSILON, ITERATIONS, SCALE_STEP = 5, 5, 0.01, 300, 0.8
GRAD_IS_ZERO = 1e-10
F = np.float32


def resize_bilinear(img: np.ndarray, h: int, w: int) -> np.ndarray:
    """cv::resize INTER_LINEAR on a float image: pixel centres map to (i + 0.5) * in / out - 0.5, edges replicate."""
    ih, iw = img.shape
    ys = np.clip((np.arange(h, dtype=np.float64) + 0.5) * ih / h - 0.5, 0, ih - 1)
    xs = np.clip((np.arange(w, dtype=np.float64) + 0.5) * iw / w - 0.5, 0, iw - 1)
    y0, x0 = np.floor(ys).astype(int), np.floor(xs).astype(int)
    y1, x1 = np.minimum(y0 + 1, ih - 1), np.minimum(x0 + 1, iw - 1)
    wy, wx = (ys - y0).astype(F)[:, None], (xs - x0).astype(F)[None, :]
    a = img.astype(F)
    top = a[y0][:, x0] * (F(1) - wx) + a[y0][:, x1] * wx
    bot = a[y1][:, x0] * (F(1) - wx) + a[y1][:, x1] * wx
    return (top * (F(1) - wy) + bot * wy).astype(F)


def pyramid_sizes(h: int, w: int, nscales: int = NSCALES, scale_step: float = SCALE_STEP) -> List[Tuple[int, int]]:
    """Level sizes, finest first: each level is round(previous * scale_step); the pyramid stops before 16 pixels."""
    sizes = [(h, w)]
    for _ in range(1, nscales):
        nh, nw = int(round(sizes[-1][0] * scale_step)), int(round(sizes[-1][1] * scale_step))
        if nh < 16 or nw < 16:
            break
        sizes.append((nh, nw))
    return sizes


def centered_gradient(img: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """Central differences 0.5 (I[x+1] - I[x-1]) in the interior, one-sided 0.5 (I[1] - I[0]) style at the border
    (IPOL centered_gradient: the missing neighbour is replaced by the pixel itself)."""
    p = np.pad(img, 1, mode="edge")
    gx = F(0.5) * (p[1:-1, 2:] - p[1:-1, :-2])
    gy = F(0.5) * (p[2:, 1:-1] - p[:-2, 1:-1])
    return gx.astype(F), gy.astype(F)


def forward_gradient(u: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """u[x+1] - u[x], zero in the last column / row."""
    ux = np.zeros_like(u)
    uy = np.zeros_like(u)
    ux[:, :-1] = u[:, 1:] - u[:, :-1]
    uy[:-1, :] = u[1:, :] - u[:-1, :]
    return ux, uy


def divergence(p1: np.ndarray, p2: np.ndarray) -> np.ndarray:
    """Backward differences, the adjoint of forward_gradient: p1[x] - p1[x-1] + p2[y] - p2[y-1], with p[-1] = 0."""
    dx = p1.copy()
    dx[:, 1:] = p1[:, 1:] - p1[:, :-1]
    dy = p2.copy()
    dy[1:, :] = p2[1:, :] - p2[:-1, :]
    return (dx + dy).astype(F)


def warp_bilinear(img: np.ndarray, u1: np.ndarray, u2: np.ndarray) -> np.ndarray:
    """img sampled at (x + u1, y + u2), bilinear, coordinates clamped to the image (border replicate)."""
    h, w = img.shape
    xs = np.clip(np.arange(w, dtype=F)[None, :] + u1, F(0), F(w - 1)).astype(F)
    ys = np.clip(np.arange(h, dtype=F)[:, None] + u2, F(0), F(h - 1)).astype(F)
    x0, y0 = np.floor(xs).astype(int), np.floor(ys).astype(int)
    x1, y1 = np.minimum(x0 + 1, w - 1), np.minimum(y0 + 1, h - 1)
    wx, wy = (xs - x0.astype(F)).astype(F), (ys - y0.astype(F)).astype(F)
    top = img[y0, x0] * (F(1) - wx) + img[y0, x1] * wx
    bot = img[y1, x0] * (F(1) - wx) + img[y1, x1] * wx
    return (top * (F(1) - wy) + bot * wy).astype(F)


def tvl1_level(i0: np.ndarray, i1: np.ndarray, u1: np.ndarray, u2: np.ndarray, warps: int = WARPS, iterations: int = ITERATIONS,
               epsilon: float = EPSILON, tau: float = TAU, lam: float = LAMBDA, theta: float = THETA):
    """One pyramid level of IPOL Algorithm 1.  Returns (u1, u2, inner iterations actually run per warp)."""
    i0, i1 = i0.astype(F), i1.astype(F)
    u1, u2 = u1.astype(F).copy(), u2.astype(F).copy()
    tau, lam, theta, epsilon = float(F(tau)), float(F(lam)), float(F(theta)), float(F(epsilon))   # the C ABI carries float32 parameters
    l_t, taut, th = F(lam * theta), F(tau / theta), F(theta)
    i1x, i1y = centered_gradient(i1)
    p11, p12, p21, p22 = (np.zeros_like(u1) for _ in range(4))
    ran = []
    for _ in range(warps):
        i1w, i1wx, i1wy = warp_bilinear(i1, u1, u2), warp_bilinear(i1x, u1, u2), warp_bilinear(i1y, u1, u2)
        grad = (i1wx * i1wx + i1wy * i1wy).astype(F)
        rho_c = (i1w - i1wx * u1 - i1wy * u2 - i0).astype(F)
        n = 0
        error = np.inf
        while error > epsilon * epsilon and n < iterations:
            n += 1
            rho = (rho_c + (i1wx * u1 + i1wy * u2)).astype(F)
            lo, hi = rho < -l_t * grad, rho > l_t * grad
            mid = ~lo & ~hi & (grad > F(GRAD_IS_ZERO))
            fi = np.where(mid, -rho / np.where(mid, grad, F(1)), F(0)).astype(F)
            d1 = np.where(lo, l_t * i1wx, np.where(hi, -l_t * i1wx, fi * i1wx)).astype(F)
            d2 = np.where(lo, l_t * i1wy, np.where(hi, -l_t * i1wy, fi * i1wy)).astype(F)
            v1, v2 = u1 + d1, u2 + d2
            n1 = (v1 + th * divergence(p11, p12)).astype(F)
            n2 = (v2 + th * divergence(p21, p22)).astype(F)
            d = (n1 - u1) * (n1 - u1) + (n2 - u2) * (n2 - u2)
            error = float(d.astype(np.float64).sum()) / d.size          # mean squared update, accumulated in fp64
            u1, u2 = n1, n2
            u1x, u1y = forward_gradient(u1)
            u2x, u2y = forward_gradient(u2)
            ng1 = F(1) + taut * np.sqrt(u1x * u1x + u1y * u1y)
            ng2 = F(1) + taut * np.sqrt(u2x * u2x + u2y * u2y)
            p11, p12 = ((p11 + taut * u1x) / ng1).astype(F), ((p12 + taut * u1y) / ng1).astype(F)
            p21, p22 = ((p21 + taut * u2x) / ng2).astype(F), ((p22 + taut * u2y) / ng2).astype(F)
        ran.append(n)
    return u1, u2, ran


def tvl1_flow(frame0: np.ndarray, frame1: np.ndarray, nscales: int = NSCALES, warps: int = WARPS, iterations: int = ITERATIONS,
              epsilon: float = EPSILON, scale_step: float = SCALE_STEP, **kw):
    """Flow (u1 = dx, u2 = dy) from grey uint8 (or float, 0..255) frame0 to frame1: coarse-to-fine over the pyramid, the
    flow of a level resized to the next finer one and divided by scale_step."""
    f0, f1 = frame0.astype(F), frame1.astype(F)
    scale_step = float(F(scale_step))
    sizes = pyramid_sizes(f0.shape[0], f0.shape[1], nscales, scale_step)
    pyr0, pyr1 = [f0], [f1]
    for (h, w) in sizes[1:]:
        pyr0.append(resize_bilinear(pyr0[-1], h, w))
        pyr1.append(resize_bilinear(pyr1[-1], h, w))
    u1 = np.zeros(sizes[-1], dtype=F)
    u2 = np.zeros(sizes[-1], dtype=F)
    counts = []
    for s in range(len(sizes) - 1, -1, -1):
        u1, u2, ran = tvl1_level(pyr0[s], pyr1[s], u1, u2, warps, iterations, epsilon, **kw)
        counts.append(ran)
        if s > 0:
            h, w = sizes[s - 1]
            inv = F(1.0 / scale_step)
            u1, u2 = (resize_bilinear(u1, h, w) * inv).astype(F), (resize_bilinear(u2, h, w) * inv).astype(F)
    return u1, u2, counts


def flow_to_image(flow: np.ndarray, bound: float = 20.0) -> np.ndarray:
    """dense_flow's convertFlowToImage with ``-b 20``: [-bound, bound] -> [0, 255], round half away from zero (cvRound of
    a non-negative value), clamped."""
    v = (flow.astype(np.float64) + bound) * (255.0 / (2.0 * bound))
    return np.clip(np.floor(v + 0.5), 0, 255).astype(np.uint8)


def warp_homography(img: np.ndarray, hmat: Sequence[Sequence[float]]) -> np.ndarray:
    """cv::warpPerspective(img, H) with INTER_LINEAR and a replicated border: out(x, y) = img(H^-1 (x, y, 1)).  The
    matrix itself (feature matches + RANSAC in the reference's binary, oracle/warp_oracle.py) is an INPUT here."""
    h, w = img.shape
    hi = np.linalg.inv(np.asarray(hmat, dtype=np.float64))
    ys, xs = np.mgrid[0:h, 0:w].astype(np.float64)
    den = hi[2, 0] * xs + hi[2, 1] * ys + hi[2, 2]
    sx = ((hi[0, 0] * xs + hi[0, 1] * ys + hi[0, 2]) / den).astype(F)
    sy = ((hi[1, 0] * xs + hi[1, 1] * ys + hi[1, 2]) / den).astype(F)
    return warp_bilinear(img.astype(F), sx - xs.astype(F), sy - ys.astype(F))
