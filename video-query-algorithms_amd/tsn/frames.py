"""Frame ingest for the TSN feature extractor: what sits between the frame directories and the network input.

Restated from the reference's driver (src/features_GPU_compute/calcSig_wOF.py) and, where the reference hands
over to the un-vendored ``pyActionRecog`` package, from that package's documented behaviour (SURVEY.md
Appendix B -- those parts are "parity unpinned"):
  * ``parse_directory``      -> {clip: path}, {clip: #rgb}, {clip: #flow}           (calcSig_wOF.py:198)
  * ``frame_ticks``          -> the T sampled frame numbers of a clip               (calcSig_wOF.py:67-72)
  * ``flow_stack_indices``   -> the 5 flow frames of a snippet                       (calcSig_wOF.py:104)
  * ``crop0``                -> resize to 340x256 and keep the top-left 224x224 crop: the only one of the 10
                                over-sampled crops whose feature the reference keeps (calcSig_wOF.py:94-95)
Image decoding on the host uses cv2 when present, else PIL, else the built-in PPM/PGM/NPY reader; ``load_*_jpegs`` hand the
undecoded JPEG files to the library's own decoder instead (tsn/jpeg.py: same pixels as libjpeg, no host image library).
"""
from __future__ import annotations

import glob
import os
from typing import Dict, List, Tuple

import numpy as np


def parse_directory(path: str, rgb_prefix='img_', flow_x_prefix='flow_x_', flow_y_prefix='flow_y_'):
    """Sub-directories of `path` are clips; count frames by prefix; x/y flow counts must agree."""
    dir_dict, rgb_counts, flow_counts = {}, {}, {}
    for d in sorted(glob.glob(os.path.join(path, '*'))):
        if not os.path.isdir(d):
            continue
        names = os.listdir(d)
        k = os.path.basename(d)
        nr = sum(1 for n in names if n.startswith(rgb_prefix))
        nx = sum(1 for n in names if n.startswith(flow_x_prefix))
        ny = sum(1 for n in names if n.startswith(flow_y_prefix))
        if nx != ny:
            raise ValueError('x and y direction have different number of flow images. video: ' + d)
        dir_dict[k], rgb_counts[k], flow_counts[k] = d, nr, nx
    return dir_dict, rgb_counts, flow_counts


def frame_ticks(frame_cnt: int, num_frame_per_video: int, stack_depth: int) -> List[int]:
    """calcSig_wOF.py:67-72 under Python-2 integer division (``T == 1`` raises ZeroDivisionError there too)."""
    step = (frame_cnt - stack_depth) // (num_frame_per_video - 1)
    if step > 0:
        frame_ticks_ = list(range(1, min((2 + step * (num_frame_per_video - 1)), frame_cnt + 1), step))
    else:
        frame_ticks_ = [1] * num_frame_per_video
    assert (len(frame_ticks_) == num_frame_per_video)
    return frame_ticks_


def flow_stack_indices(tick: int, frame_cnt: int, stk_depth: int) -> List[int]:
    return [min(frame_cnt, tick + offset) for offset in range(stk_depth)]


# ------------------------------------------------------------------------------------------------ decoding
def clip_plan(n_frames: int, frames_per_clip: int = 150, frames_per_second: int = 15):
    """How ``create_clip`` cuts a video's frames into clips (src/features_GPU_compute/build_wof_clips.py:78-128):
    ``int(n / frames_per_clip)`` full clips of consecutive frames (1-based), then the remaining frames become one more,
    shorter clip if they last at least 2 seconds, else they are deleted.  Returns [(clip number, first frame, last frame)]
    and the number of frames dropped.  (The flow frames themselves: tsn/flow.py; the whole command line: build_wof_clips.py.)"""
    nclips = int(n_frames / frames_per_clip)
    plan = [(n + 1, n * frames_per_clip + 1, (n + 1) * frames_per_clip) for n in range(nclips)]
    remaining = n_frames - nclips * frames_per_clip
    if remaining >= 2 * frames_per_second:
        plan.append((nclips + 1, nclips * frames_per_clip + 1, nclips * frames_per_clip + remaining))
        return plan, 0
    return plan, remaining


def _read_pnm(path: str) -> np.ndarray:
    with open(path, 'rb') as f:
        data = f.read()
    parts, pos = [], 0
    while len(parts) < 4:
        while data[pos:pos + 1].isspace():
            pos += 1
        if data[pos:pos + 1] == b'#':
            pos = data.index(b'\n', pos) + 1
            continue
        end = pos
        while not data[end:end + 1].isspace():
            end += 1
        parts.append(data[pos:end])
        pos = end
    magic, w, h = parts[0], int(parts[1]), int(parts[2])
    pos += 1
    if magic == b'P6':
        return np.frombuffer(data, dtype=np.uint8, count=w * h * 3, offset=pos).reshape(h, w, 3)[:, :, ::-1].copy()   # -> BGR
    if magic == b'P5':
        return np.frombuffer(data, dtype=np.uint8, count=w * h, offset=pos).reshape(h, w).copy()
    raise ValueError("unsupported PNM file " + path)


def imread(path: str, color: bool) -> np.ndarray:
    """BGR uint8 [H,W,3] (cv2.IMREAD_COLOR order) or grayscale uint8 [H,W]."""
    ext = os.path.splitext(path)[1].lower()
    if ext == '.npy':
        return np.load(path, allow_pickle=False)
    if ext in ('.ppm', '.pgm', '.pnm'):
        return _read_pnm(path)
    try:
        import cv2
        img = cv2.imread(path, cv2.IMREAD_COLOR if color else cv2.IMREAD_GRAYSCALE)
        if img is None:
            raise IOError("cannot read " + path)
        return img
    except ImportError:
        pass
    try:
        from PIL import Image
        im = Image.open(path)
        if color:
            return np.asarray(im.convert('RGB'))[:, :, ::-1].copy()
        if im.format == 'JPEG' and im.mode != 'L':
            im.draft('L', im.size)      # libjpeg's own grayscale output (the Y plane), as cv2.IMREAD_GRAYSCALE gets it
        return np.asarray(im.convert('L'))
    except ImportError:
        raise ImportError("no JPEG decoder in this environment (cv2 / PIL missing); use .ppm/.pgm/.npy frames "
                          "(--frame_ext) or install one")


def write_pnm(path: str, img: np.ndarray):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    with open(path, 'wb') as f:
        if img.ndim == 3:
            f.write(b'P6\n%d %d\n255\n' % (img.shape[1], img.shape[0]))
            f.write(img[:, :, ::-1].tobytes())
        else:
            f.write(b'P5\n%d %d\n255\n' % (img.shape[1], img.shape[0]))
            f.write(img.tobytes())


RESIZE_CV2, RESIZE_EXACT = 0, 1            # include/vq_amd.h: VQ_RESIZE_CV2_FIXED / VQ_RESIZE_EXACT
RESIZE_RULES = {"cv2": RESIZE_CV2, "exact": RESIZE_EXACT}


def _cv2_linear_taps(n_in: int, n_out: int, clamp_taps: bool):
    """Source index and the two 11-bit weights of every output position under cv::resize's INTER_LINEAR rule for 8-bit
    images: ``scale = 1 / (n_out / n_in)`` in double, ``f = (float)((d + 0.5) * scale - 0.5)``, ``s = floor(f)``, ``f -= s``
    in float, weights ``cvRound((1.f - f) * 2048)`` and ``cvRound(f * 2048)`` (round half to even) as int16.  Along x a tap
    left of the image or on its last column is moved onto the border pixel with f = 0 (``clamp_taps``); along y the weights
    stay and the two ROWS are clipped to the image instead."""
    scale = 1.0 / (float(n_out) / float(n_in))
    f = ((np.arange(n_out, dtype=np.float64) + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    if clamp_taps:
        low, high = s < 0, s >= n_in - 1
        f = np.where(low | high, np.float32(0), f).astype(np.float32)
        s = np.where(low, 0, np.where(high, n_in - 1, s))
    w0 = np.rint((np.float32(1) - f) * np.float32(2048)).astype(np.int64)
    w1 = np.rint(f * np.float32(2048)).astype(np.int64)
    return s, w0, w1


def resize_cv2_fixed(img: np.ndarray, size_wh: Tuple[int, int]) -> np.ndarray:
    """``cv2.resize(img, size_wh)`` (INTER_LINEAR) for uint8 in integer arithmetic, as OpenCV's generic resize does it:
    horizontal pass ``S = p[sx] * a0 + p[sx + 1] * a1`` (int32, weights scaled by 2^11), vertical pass
    ``(((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2``.  Restated from memory of imgproc/resize.cpp (cv2 is
    not in this image; parity unpinned); host, device (csrc/vq_frames.hip) and oracle agree bit for bit."""
    w, h = size_wh
    ih, iw = img.shape[:2]
    sx, a0, a1 = _cv2_linear_taps(iw, w, True)
    sy, b0, b1 = _cv2_linear_taps(ih, h, False)
    x1 = np.minimum(sx + 1, iw - 1)
    y0, y1 = np.clip(sy, 0, ih - 1), np.clip(sy + 1, 0, ih - 1)
    a = img.astype(np.int64)
    if a.ndim == 3:
        a0, a1 = a0[:, None], a1[:, None]
    rows = a[:, sx] * a0 + a[:, x1] * a1                           # [ih, w(, c)]: the horizontal pass of every source row
    bshape = (-1, 1) + (1,) * (a.ndim - 2)
    out = (((b0.reshape(bshape) * (rows[y0] >> 4)) >> 16) + ((b1.reshape(bshape) * (rows[y1] >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


def resize_exact(img: np.ndarray, size_wh: Tuple[int, int]) -> np.ndarray:
    """Bilinear resize on cv2.resize's sampling grid (half-pixel centres) with exact fp64 weights, rounded half to even --
    the rule of rounds 1-2, kept as an option (``--exact_resize``): differs from the fixed-point rule by at most one grey
    level on about one pixel in eight."""
    w, h = size_wh
    ih, iw = img.shape[:2]
    ys = np.clip((np.arange(h) + 0.5) * ih / h - 0.5, 0, ih - 1)
    xs = np.clip((np.arange(w) + 0.5) * iw / w - 0.5, 0, iw - 1)
    y0, x0 = np.floor(ys).astype(int), np.floor(xs).astype(int)
    y1, x1 = np.minimum(y0 + 1, ih - 1), np.minimum(x0 + 1, iw - 1)
    wy, wx = (ys - y0)[:, None], (xs - x0)[None, :]
    a = img.astype(np.float64)
    if a.ndim == 3:
        wy, wx = wy[..., None], wx[..., None]
    out = (a[y0][:, x0] * (1 - wy) * (1 - wx) + a[y0][:, x1] * (1 - wy) * wx
           + a[y1][:, x0] * wy * (1 - wx) + a[y1][:, x1] * wy * wx)
    return np.clip(np.rint(out), 0, 255).astype(np.uint8)


def resize_bilinear(img: np.ndarray, size_wh: Tuple[int, int], rule: str = "cv2") -> np.ndarray:
    """The ``cv2.resize(frame, size_wh)`` of the reference's dependency (calcSig_wOF.py:94,111 through
    predict_single_frame / predict_single_flow_stack; build_wof_clips.py:41): rule "cv2" = OpenCV's 11-bit fixed-point
    INTER_LINEAR (the default: a drop-in follows the dependency), "exact" = exact fp64 weights.  An image that already has
    the size is returned as it is (cv::resize copies it)."""
    w, h = size_wh
    if img.shape[:2] == (h, w):
        return img
    if rule not in RESIZE_RULES:
        raise ValueError("resize rule must be 'cv2' or 'exact', not %r" % (rule,))
    return resize_cv2_fixed(img, size_wh) if rule == "cv2" else resize_exact(img, size_wh)


def crop0(img: np.ndarray, frame_size=(340, 256), crop=224, rule: str = "cv2") -> np.ndarray:
    """Resize to frame_size (w, h) and keep over-sample crop 0 = top-left, un-mirrored."""
    return resize_bilinear(img, frame_size, rule)[:crop, :crop]


def load_rgb_frames(clip_dir: str, ticks: List[int], rgb_prefix='img_', ext='.jpg') -> np.ndarray:
    """[T, H, W, 3] uint8 BGR: the decoded frames of the snippets, NOT resized (the device does that: vq_resize_crop)."""
    return np.stack([imread(os.path.join(clip_dir, '{}{:05d}{}'.format(rgb_prefix, t, ext)), True) for t in ticks])


def load_flow_frames(clip_dir: str, ticks: List[int], frame_cnt: int, stk_depth=5, flow_x_prefix='flow_x_',
                     flow_y_prefix='flow_y_', ext='.jpg') -> np.ndarray:
    """[T, 2*stk_depth, H, W] uint8: the x/y flow frames of every snippet in stack order (x0, y0, x1, y1, ...), NOT resized."""
    out = []
    for tick in ticks:
        planes = []
        for idx in flow_stack_indices(tick, frame_cnt, stk_depth):
            planes.append(imread(os.path.join(clip_dir, '{}{:05d}{}'.format(flow_x_prefix, idx, ext)), False))
            planes.append(imread(os.path.join(clip_dir, '{}{:05d}{}'.format(flow_y_prefix, idx, ext)), False))
        out.append(np.stack(planes))
    return np.stack(out)


def _read_file(path: str) -> bytes:
    with open(path, 'rb') as f:
        return f.read()


def load_rgb_jpegs(clip_dir: str, ticks: List[int], rgb_prefix='img_', ext='.jpg') -> List[str]:
    """The JPEG files of the snippets as PATHS: the library reads and decodes them (tsn/jpeg.py)."""
    return [os.path.join(clip_dir, '{}{:05d}{}'.format(rgb_prefix, t, ext)) for t in ticks]


def load_flow_jpegs(clip_dir: str, ticks: List[int], frame_cnt: int, stk_depth=5, flow_x_prefix='flow_x_', flow_y_prefix='flow_y_',
                    ext='.jpg') -> List[str]:
    """The x / y flow JPEG files of every snippet in stack order (x0, y0, x1, y1, ...) as PATHS.  (Neighbouring snippets share most of
    their frames -- 250 names of ~30 files per clip at the reference's defaults --, so a name is formatted once per clip: 8 ms of
    interpreter time per batch of 32 clips became 3.)"""
    names = {}
    out = []
    for tick in ticks:
        for idx in flow_stack_indices(tick, frame_cnt, stk_depth):
            pair = names.get(idx)
            if pair is None:
                pair = names[idx] = (os.path.join(clip_dir, '{}{:05d}{}'.format(flow_x_prefix, idx, ext)),
                                     os.path.join(clip_dir, '{}{:05d}{}'.format(flow_y_prefix, idx, ext)))
            out += pair
    return out


def load_rgb_snippets(clip_dir: str, ticks: List[int], rgb_prefix='img_', ext='.jpg', rule: str = "cv2") -> np.ndarray:
    """[T, 224, 224, 3] uint8 BGR (calcSig_wOF.py:88-96)."""
    return np.stack([crop0(imread(os.path.join(clip_dir, '{}{:05d}{}'.format(rgb_prefix, t, ext)), True), rule=rule) for t in ticks])


def load_flow_snippets(clip_dir: str, ticks: List[int], frame_cnt: int, stk_depth=5, flow_x_prefix='flow_x_',
                       flow_y_prefix='flow_y_', ext='.jpg', rule: str = "cv2") -> np.ndarray:
    """[T, 224, 224, 10] uint8: x/y of 5 consecutive flow frames interleaved (calcSig_wOF.py:99-113)."""
    out = []
    for tick in ticks:
        stack = []
        for idx in flow_stack_indices(tick, frame_cnt, stk_depth):
            stack.append(crop0(imread(os.path.join(clip_dir, '{}{:05d}{}'.format(flow_x_prefix, idx, ext)), False), rule=rule))
            stack.append(crop0(imread(os.path.join(clip_dir, '{}{:05d}{}'.format(flow_y_prefix, idx, ext)), False), rule=rule))
        out.append(np.stack(stack, axis=-1))
    return np.stack(out)
