"""TEST INFRASTRUCTURE (oracle) -- frame ingest in front of the TSN forward, restated pixel by pixel.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file; the product never does.

What it restates: the resize + over-sample step the reference reaches through
``net.predict_single_frame([frame], score_name, frame_size=(340, 256))`` (src/features_GPU_compute/calcSig_wOF.py:94)
and ``net.predict_single_flow_stack(flow_stack, score_name, frame_size=(340, 256))`` (:111), of which it keeps crop 0
only (``.data[0]``, :95, :112).  The arithmetic lives in the un-vendored third-party ``pyActionRecog`` (yjxiong/
temporal-segment-networks, no pinned commit; SURVEY.md Appendix B, from memory): ``cv2.resize(frame, (340, 256))``
(INTER_LINEAR: output pixel centres map to ``(i + 0.5) * in / out - 0.5``, clamped to the image) followed by the 10-crop
over-sample whose crop 0 is the top-left 224 x 224 window, un-mirrored.

PARITY UNPINNED: neither cv2 nor any frame of the reference exists here.  cv2's uint8 path evaluates the bilinear weights
in 11-bit fixed point: ``fixed_point_resize_pixel`` states that rule as remembered from OpenCV's imgproc/resize.cpp
(resizeGeneric_ / HResizeLinear / VResizeLinear with FixedPtCast<int, uchar, 22>), and it is the rule the product follows by
default since round 3 (rule "cv2"); ``resize_pixel`` is the exact-weight rule of rounds 1-2, kept as the product's
``--exact_resize`` option.  The two differ by at most one grey level (about one pixel in eight).

Written as plain Python scalar loops on purpose: it shares no code and no vectorisation strategy with
tsn/frames.py (numpy fancy indexing) or csrc/vq_frames.hip (one thread per output pixel).
"""
from __future__ import annotations

import math
import struct

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


def crop0(img: np.ndarray, frame_size=(340, 256), crop: int = 224, rule: str = "cv2") -> np.ndarray:
    """Resize to frame_size = (w, h), keep the top-left crop x crop window (over-sample crop 0).  Only the pixels that
    survive the crop are evaluated.  An image that already has the frame size is passed through (cv::resize copies it).
    rule "cv2": OpenCV's fixed-point rule (the product's default); "exact": exact fp64 weights."""
    out_w, out_h = frame_size
    rows = img.tolist()
    grey = img.ndim == 2
    if img.shape[:2] == (out_h, out_w):
        return np.array([r[:crop] for r in rows[:crop]], dtype=np.uint8)
    pixel = {"cv2": fixed_point_resize_pixel, "exact": resize_pixel}[rule]
    out = []
    for y in range(crop):
        line = []
        for x in range(crop):
            if grey:
                line.append(pixel(rows, y, x, out_h, out_w, None))
            else:
                line.append([pixel(rows, y, x, out_h, out_w, c) for c in range(img.shape[2])])
        out.append(line)
    return np.array(out, dtype=np.uint8)


def flow_stack_crop0(planes, frame_size=(340, 256), crop: int = 224, rule: str = "cv2") -> np.ndarray:
    """[x0, y0, x1, y1, ...] grey flow frames of one snippet (calcSig_wOF.py:104-110) -> [crop][crop][2 * depth]."""
    return np.stack([crop0(p, frame_size, crop, rule) for p in planes], axis=-1)


def _f32(v) -> float:
    """v rounded to IEEE binary32 (as a Python float): every ``float`` variable of the C++ code goes through this."""
    return struct.unpack("f", struct.pack("f", v))[0]


def _cv_round(v: float) -> int:
    """cvRound: round to nearest, ties to even (SSE cvtss2si / lrint under the default rounding mode)."""
    return int(round(v))            # Python's round() is round-half-to-even


def _linear_tap(d: int, n_in: int, n_out: int, clamp_tap: bool):
    """OpenCV's resize(): ``double scale = 1. / ((double)n_out / n_in); float f = (float)((d + 0.5) * scale - 0.5);
    int s = cvFloor(f); f -= s;`` -- and, for the x axis only, a tap left of the image or on its last column moves onto the
    border pixel with f = 0.  Weights: ``saturate_cast<short>(cbuf[k] * INTER_RESIZE_COEF_SCALE)`` with cbuf = {1.f - f, f}."""
    scale = 1.0 / (float(n_out) / float(n_in))
    f = _f32((d + 0.5) * scale - 0.5)
    s = int(math.floor(f))
    f = _f32(f - float(s))
    if clamp_tap:
        if s < 0:
            s, f = 0, 0.0
        if s >= n_in - 1:
            s, f = n_in - 1, 0.0
    w0 = _cv_round(_f32(_f32(1.0 - f) * 2048.0))
    w1 = _cv_round(_f32(f * 2048.0))
    return s, w0, w1


def fixed_point_resize_pixel(img, y: int, x: int, out_h: int, out_w: int, ch):
    """One output value under cv2's uint8 INTER_LINEAR rule AS REMEMBERED (OpenCV imgproc/resize.cpp: weights scaled by
    2^11 and rounded to int16; horizontal pass ``S = p[sx] * a0 + p[sx + 1] * a1`` in int32; vertical pass over the two
    source rows CLIPPED to the image, ``(((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2``).  Unverified here
    (no cv2, no reference frames): parity unpinned."""
    in_h, in_w = len(img), len(img[0])
    sx, a0, a1 = _linear_tap(x, in_w, out_w, True)
    sy, b0, b1 = _linear_tap(y, in_h, out_h, False)
    x1 = min(sx + 1, in_w - 1)
    y0, y1 = min(max(sy, 0), in_h - 1), min(max(sy + 1, 0), in_h - 1)

    def px(r, c):
        return int(img[r][c] if ch is None else img[r][c][ch])
    s0 = px(y0, sx) * a0 + px(y0, x1) * a1
    s1 = px(y1, sx) * a0 + px(y1, x1) * a1
    v = (((b0 * (s0 >> 4)) >> 16) + ((b1 * (s1 >> 4)) >> 16) + 2) >> 2
    return min(max(v, 0), 255)
