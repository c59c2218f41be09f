"""TEST INFRASTRUCTURE (oracle) -- frame ingest in front of the TSN forward, restated pixel by pixel.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file; the product never does.

What it restates: the resize + over-sample step the reference reaches through
``net.predict_single_frame([frame], score_name, frame_size=(340, 256))`` (src/features_GPU_compute/calcSig_wOF.py:94)
and ``net.predict_single_flow_stack(flow_stack, score_name, frame_size=(340, 256))`` (:111), of which it keeps crop 0
only (``.data[0]``, :95, :112).  The arithmetic lives in the un-vendored third-party ``pyActionRecog`` (yjxiong/
temporal-segment-networks, no pinned commit; SURVEY.md Appendix B, from memory): ``cv2.resize(frame, (340, 256))``
(INTER_LINEAR: output pixel centres map to ``(i + 0.5) * in / out - 0.5``, clamped to the image) followed by the 10-crop
over-sample whose crop 0 is the top-left 224 x 224 window, un-mirrored.

PARITY UNPINNED: neither cv2 nor any frame of the reference exists here.  In particular cv2's uint8 path evaluates the
bilinear weights in 11-bit fixed point (``fixed_point_resize_pixel`` below states that rule as remembered); the product
uses the exact fp64 weights (``resize_pixel``), which can differ from it by one grey level on some pixels.

Written as plain Python scalar loops on purpose: it shares no code and no vectorisation strategy with
tsn/frames.py (numpy fancy indexing) or csrc/vq_frames.hip (one thread per output pixel).
"""
from __future__ import annotations

import math

import numpy as np


def source_coordinate(i: int, n_in: int, n_out: int) -> float:
    """Where output pixel centre i falls in the input (half-pixel centres), clamped to the image."""
    return min(max((i + 0.5) * n_in / n_out - 0.5, 0.0), float(n_in - 1))


def resize_pixel(img, y: int, x: int, out_h: int, out_w: int, ch):
    """One output value of the bilinear resize of img [H][W](C) to out_h x out_w, fp64, rounded half to even."""
    in_h, in_w = len(img), len(img[0])
    ys, xs = source_coordinate(y, in_h, out_h), source_coordinate(x, in_w, out_w)
    y0, x0 = int(math.floor(ys)), int(math.floor(xs))
    y1, x1 = min(y0 + 1, in_h - 1), min(x0 + 1, in_w - 1)
    wy, wx = ys - y0, xs - x0

    def px(r, c):
        return float(img[r][c] if ch is None else img[r][c][ch])
    v = px(y0, x0) * (1 - wy) * (1 - wx)
    v = v + px(y0, x1) * (1 - wy) * wx
    v = v + px(y1, x0) * wy * (1 - wx)
    v = v + px(y1, x1) * wy * wx
    return int(min(max(round(v), 0), 255))          # Python's round() is round-half-to-even, like numpy.rint


def crop0(img: np.ndarray, frame_size=(340, 256), crop: int = 224) -> np.ndarray:
    """Resize to frame_size = (w, h), keep the top-left crop x crop window (over-sample crop 0).  Only the pixels that
    survive the crop are evaluated.  An image that already has the frame size is passed through (cv2.resize would
    return the same pixels)."""
    out_w, out_h = frame_size
    rows = img.tolist()
    grey = img.ndim == 2
    if img.shape[:2] == (out_h, out_w):
        return np.array([r[:crop] for r in rows[:crop]], dtype=np.uint8)
    out = []
    for y in range(crop):
        line = []
        for x in range(crop):
            if grey:
                line.append(resize_pixel(rows, y, x, out_h, out_w, None))
            else:
                line.append([resize_pixel(rows, y, x, out_h, out_w, c) for c in range(img.shape[2])])
        out.append(line)
    return np.array(out, dtype=np.uint8)


def flow_stack_crop0(planes, frame_size=(340, 256), crop: int = 224) -> np.ndarray:
    """[x0, y0, x1, y1, ...] grey flow frames of one snippet (calcSig_wOF.py:104-110) -> [crop][crop][2 * depth]."""
    return np.stack([crop0(p, frame_size, crop) for p in planes], axis=-1)


def fixed_point_resize_pixel(img, y: int, x: int, out_h: int, out_w: int, ch):
    """The same output value under cv2's uint8 INTER_LINEAR rule AS REMEMBERED (OpenCV imgproc resize.cpp: weights
    scaled by 2^11 and rounded to int16; horizontal pass in int32; vertical pass
    ``(((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2``).  Unverified here; kept to measure how far the
    exact-weight path can be from it (tests report the maximum difference; it is not asserted to be zero)."""
    in_h, in_w = len(img), len(img[0])
    fy = (y + 0.5) * in_h / out_h - 0.5
    fx = (x + 0.5) * in_w / out_w - 0.5
    sy, sx = int(math.floor(fy)), int(math.floor(fx))
    fy, fx = fy - sy, fx - sx
    if sy < 0:
        sy, fy = 0, 0.0
    if sy >= in_h - 1:
        sy, fy = in_h - 1, 0.0
    if sx < 0:
        sx, fx = 0, 0.0
    if sx >= in_w - 1:
        sx, fx = in_w - 1, 0.0
    scale = 1 << 11
    a1 = int(round(fx * scale))
    a0 = scale - a1
    b1 = int(round(fy * scale))
    b0 = scale - b1
    y1, x1 = min(sy + 1, in_h - 1), min(sx + 1, in_w - 1)

    def px(r, c):
        return int(img[r][c] if ch is None else img[r][c][ch])
    s0 = px(sy, sx) * a0 + px(sy, x1) * a1
    s1 = px(y1, sx) * a0 + px(y1, x1) * a1
    v = (((b0 * (s0 >> 4)) >> 16) + ((b1 * (s1 >> 4)) >> 16) + 2) >> 2
    return min(max(v, 0), 255)
