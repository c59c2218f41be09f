"""TEST INFRASTRUCTURE (oracle) -- baseline JPEG decoding, the first step of the reference's frame ingest.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file; the product never does.

What the reference does: ``cv2.imread(path, cv2.IMREAD_COLOR)`` for RGB frames and ``cv2.imread(path,
cv2.IMREAD_GRAYSCALE)`` for the flow images (src/features_GPU_compute/calcSig_wOF.py:92,105-106); the files are the
``img_NNNNN.jpg`` / ``flow_x_NNNNN.jpg`` / ``flow_y_NNNNN.jpg`` that build_wof_clips.py:46,70-73 writes (cv2.imwrite and the
dense_flow binary: baseline JPEG, 4:2:0 for colour, one component for the flow images).  cv2 decodes with libjpeg
(-turbo) at its defaults: the slow-but-accurate integer IDCT (``JDCT_ISLOW``), "fancy" (triangle-filter) chroma
upsampling, fixed-point YCbCr -> RGB.  This file restates exactly that pipeline from the published sources -- ITU-T T.81
(marker syntax, Huffman decoding, F.2.2) and the Independent JPEG Group's jidctint.c / jdsample.c / jdcolor.c algorithms
(13-bit constants, two-pass descale, (3,1)/4 and (9,3,3,1)/16 filters with their alternating rounding, 16-bit colour
tables) -- in scalar Python and numpy.

PINNED: cv2 is not installed here, but Pillow is, and Pillow binds the same libjpeg-turbo: tests/test_jpeg_oracle.py
checks this restatement against ``PIL.Image.open(...)`` bit for bit on JPEGs of odd sizes, all three chroma layouts,
several qualities and with restart markers.  (A grey read of a COLOUR file is the Y plane, as libjpeg's
``out_color_space = JCS_GRAYSCALE`` gives cv2 -- Pillow's ``convert('L')`` is a different formula and is not used.)
"""
from __future__ import annotations

import struct
from typing import Dict, List, Tuple

import numpy as np

ZIGZAG = [0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
          35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63]


class JpegError(ValueError):
    pass


def parse(data: bytes) -> Dict:
    """Marker segments of a baseline (SOF0) / extended-sequential 8-bit (SOF1) file with one scan."""
    if data[:2] != b"\xff\xd8":
        raise JpegError("not a JPEG (no SOI)")
    pos = 2
    info: Dict = {"qt": {}, "dc": {}, "ac": {}, "ri": 0}
    while True:
        while data[pos] != 0xFF:
            pos += 1
        while data[pos] == 0xFF:
            pos += 1
        m = data[pos]
        pos += 1
        if m == 0xD9:
            raise JpegError("EOI before SOS")
        ln = struct.unpack(">H", data[pos:pos + 2])[0]
        seg = data[pos + 2:pos + ln]
        pos += ln
        if m == 0xDB:
            q = 0
            while q < len(seg):
                prec, tid = seg[q] >> 4, seg[q] & 15
                q += 1
                if prec:
                    vals = struct.unpack(">64H", seg[q:q + 128])
                    q += 128
                else:
                    vals = seg[q:q + 64]
                    q += 64
                tab = [0] * 64
                for k in range(64):
                    tab[ZIGZAG[k]] = vals[k]
                info["qt"][tid] = tab
        elif m in (0xC0, 0xC1):
            prec, h, w, nc = struct.unpack(">BHHB", seg[:6])
            if prec != 8:
                raise JpegError("only 8-bit samples")
            info["h"], info["w"] = h, w
            info["comps"] = [{"id": seg[6 + 3 * i], "h": seg[7 + 3 * i] >> 4, "v": seg[7 + 3 * i] & 15, "tq": seg[8 + 3 * i]} for i in range(nc)]
        elif m in (0xC2, 0xC3, 0xC5, 0xC6, 0xC7, 0xC9, 0xCA, 0xCB, 0xCD, 0xCE, 0xCF):
            raise JpegError("unsupported JPEG process (marker %02X): baseline / sequential Huffman only" % m)
        elif m == 0xC4:
            q = 0
            while q < len(seg):
                tc, th = seg[q] >> 4, seg[q] & 15
                counts = list(seg[q + 1:q + 17])
                n = sum(counts)
                info["ac" if tc else "dc"][th] = (counts, list(seg[q + 17:q + 17 + n]))
                q += 17 + n
        elif m == 0xDD:
            info["ri"] = struct.unpack(">H", seg[:2])[0]
        elif m == 0xDA:
            ns = seg[0]
            if ns != len(info["comps"]):
                raise JpegError("multi-scan files are not supported")
            for i in range(ns):
                cid, t = seg[1 + 2 * i], seg[2 + 2 * i]
                c = next(c for c in info["comps"] if c["id"] == cid)
                c["td"], c["ta"] = t >> 4, t & 15
            info["scan"] = pos
            return info
        # APPn, COM, DNL and anything else with a length: skipped


class _Bits:
    def __init__(self, data: bytes, pos: int):
        self.d, self.p, self.acc, self.n = data, pos, 0, 0

    def _fill(self):
        b = self.d[self.p]
        if b == 0xFF:
            nxt = self.d[self.p + 1]
            if nxt == 0:
                self.p += 2
            else:                      # a marker inside the entropy-coded segment: feed zeros (T.81 F.2.2.5)
                b = 0
        else:
            self.p += 1
        self.acc = (self.acc << 8) | b
        self.n += 8

    def get(self, k: int) -> int:
        while self.n < k:
            self._fill()
        self.n -= k
        return (self.acc >> self.n) & ((1 << k) - 1)

    def restart(self) -> int:
        self.acc = self.n = 0          # discard the padding bits
        while not (self.d[self.p] == 0xFF and 0xD0 <= self.d[self.p + 1] <= 0xD7):
            self.p += 1
        m = self.d[self.p + 1]
        self.p += 2
        return m


def _huff(counts: List[int], symbols: List[int]):
    """T.81 C.2 / F.2.2.3: (mincode, maxcode, valptr) per length."""
    code, k = 0, 0
    table = {}
    for ln in range(1, 17):
        for _ in range(counts[ln - 1]):
            table[(ln, code)] = symbols[k]
            code += 1
            k += 1
        code <<= 1
    return table


def _decode_symbol(bits: _Bits, table) -> int:
    code = 0
    for ln in range(1, 17):
        code = (code << 1) | bits.get(1)
        if (ln, code) in table:
            return table[(ln, code)]
    raise JpegError("bad Huffman code")


def _extend(v: int, s: int) -> int:
    return v if v >= (1 << (s - 1)) else v - (1 << s) + 1


def decode_coefficients(data: bytes):
    """-> (info, [per component int32 array [blocks_h, blocks_w, 64] in natural order, NOT dequantised])."""
    info = parse(data)
    comps = info["comps"]
    hmax, vmax = max(c["h"] for c in comps), max(c["v"] for c in comps)
    H, W = info["h"], info["w"]
    dc = {k: _huff(*v) for k, v in info["dc"].items()}
    ac = {k: _huff(*v) for k, v in info["ac"].items()}
    single = len(comps) == 1
    if single:
        c = comps[0]
        mx, my = -(-W // 8), -(-H // 8)                                  # a one-component scan: 8x8 blocks, no MCU padding
        c["bw"], c["bh"], bh_eff, bv_eff = mx, my, 1, 1
    else:
        mx, my = -(-W // (8 * hmax)), -(-H // (8 * vmax))
        for c in comps:
            c["bw"], c["bh"] = mx * c["h"], my * c["v"]
    coef = [np.zeros((c["bh"], c["bw"], 64), np.int32) for c in comps]
    bits = _Bits(data, info["scan"])
    pred = [0] * len(comps)
    count = 0
    for mcu in range(mx * my):
        if info["ri"] and count == info["ri"]:
            bits.restart()
            pred = [0] * len(comps)
            count = 0
        count += 1
        my_, mx_ = divmod(mcu, mx)
        for ci, c in enumerate(comps):
            hh, vv = (1, 1) if single else (c["h"], c["v"])
            for by in range(vv):
                for bx in range(hh):
                    blk = coef[ci][my_ * vv + by, mx_ * hh + bx]
                    s = _decode_symbol(bits, dc[c["td"]])
                    diff = _extend(bits.get(s), s) if s else 0
                    pred[ci] += diff
                    blk[0] = pred[ci]
                    k = 1
                    while k < 64:
                        rs = _decode_symbol(bits, ac[c["ta"]])
                        r, s = rs >> 4, rs & 15
                        if s == 0:
                            if r == 15:
                                k += 16
                                continue
                            break
                        k += r
                        blk[ZIGZAG[k]] = _extend(bits.get(s), s)
                        k += 1
    info["hmax"], info["vmax"] = hmax, vmax
    return info, coef


# jidctint.c: 13-bit fixed-point constants
_F = {"0_298631336": 2446, "0_390180644": 3196, "0_541196100": 4433, "0_765366865": 6270, "0_899976223": 7373, "1_175875602": 9633,
      "1_501321110": 12299, "1_847759065": 15137, "1_961570560": 16069, "2_053119869": 16819, "2_562915447": 20995, "3_072711026": 25172}
CONST_BITS, PASS1_BITS = 13, 2


def _descale(x, n):
    return (x + (1 << (n - 1))) >> n


def _idct_1d(v, shift_in_dc):
    """One pass of jpeg_idct_islow over the LAST axis of v ([..., 8] int64)."""
    z2, z3 = v[..., 2], v[..., 6]
    z1 = (z2 + z3) * _F["0_541196100"]
    tmp2 = z1 + z3 * (-_F["1_847759065"])
    tmp3 = z1 + z2 * _F["0_765366865"]
    z2, z3 = v[..., 0], v[..., 4]
    tmp0, tmp1 = (z2 + z3) << CONST_BITS, (z2 - z3) << CONST_BITS
    tmp10, tmp13, tmp11, tmp12 = tmp0 + tmp3, tmp0 - tmp3, tmp1 + tmp2, tmp1 - tmp2
    tmp0, tmp1, tmp2, tmp3 = v[..., 7], v[..., 5], v[..., 3], v[..., 1]
    z1, z2, z3, z4 = tmp0 + tmp3, tmp1 + tmp2, tmp0 + tmp2, tmp1 + tmp3
    z5 = (z3 + z4) * _F["1_175875602"]
    tmp0, tmp1 = tmp0 * _F["0_298631336"], tmp1 * _F["2_053119869"]
    tmp2, tmp3 = tmp2 * _F["3_072711026"], tmp3 * _F["1_501321110"]
    z1, z2 = z1 * (-_F["0_899976223"]), z2 * (-_F["2_562915447"])
    z3, z4 = z3 * (-_F["1_961570560"]) + z5, z4 * (-_F["0_390180644"]) + z5
    tmp0, tmp1, tmp2, tmp3 = tmp0 + z1 + z3, tmp1 + z2 + z4, tmp2 + z2 + z3, tmp3 + z1 + z4
    n = shift_in_dc
    return np.stack([_descale(tmp10 + tmp3, n), _descale(tmp11 + tmp2, n), _descale(tmp12 + tmp1, n), _descale(tmp13 + tmp0, n),
                     _descale(tmp13 - tmp0, n), _descale(tmp12 - tmp1, n), _descale(tmp11 - tmp2, n), _descale(tmp10 - tmp3, n)], axis=-1)


def idct_blocks(coef: np.ndarray, qt: List[int]) -> np.ndarray:
    """[bh, bw, 64] quantised coefficients -> uint8 samples [bh * 8, bw * 8] (dequantise, islow IDCT, +128, clamp)."""
    bh, bw, _ = coef.shape
    x = (coef.astype(np.int64) * np.array(qt, np.int64)).reshape(bh, bw, 8, 8)          # [row u][col v]
    ws = _idct_1d(np.swapaxes(x, -1, -2), CONST_BITS - PASS1_BITS)                       # pass 1 runs down the columns
    ws = np.swapaxes(ws, -1, -2)
    out = _idct_1d(ws, CONST_BITS + PASS1_BITS + 3)                                      # pass 2 along the rows
    out = np.clip(out + 128, 0, 255).astype(np.uint8)
    return out.transpose(0, 2, 1, 3).reshape(bh * 8, bw * 8)


def upsample_h2v1(p: np.ndarray) -> np.ndarray:
    """jdsample.c h2v1_fancy_upsample: [(3,1)/4 with +1 / +2 rounding], the edge columns copied."""
    p = p.astype(np.int32)
    h, w = p.shape
    out = np.empty((h, 2 * w), np.int32)
    if w == 1:
        out[:, 0] = out[:, 1] = p[:, 0]
        return out.astype(np.uint8)
    left = np.concatenate([p[:, :1], p[:, :-1]], axis=1)
    right = np.concatenate([p[:, 1:], p[:, -1:]], axis=1)
    out[:, 0::2] = (3 * p + left + 1) >> 2
    out[:, 1::2] = (3 * p + right + 2) >> 2
    out[:, 0] = p[:, 0]
    out[:, -1] = p[:, -1]
    return out.astype(np.uint8)


def upsample_h2v2(p: np.ndarray) -> np.ndarray:
    """jdsample.c h2v2_fancy_upsample: vertical 3:1 sums of the nearer / farther row (edge rows replicated), then the
    horizontal (3,1) filter on those sums with +8 / +7 rounding and a shift by 4; first / last column from the sum alone."""
    p = p.astype(np.int32)
    h, w = p.shape
    above = np.concatenate([p[:1], p[:-1]], axis=0)
    below = np.concatenate([p[1:], p[-1:]], axis=0)
    out = np.empty((2 * h, 2 * w), np.int32)
    for v, other in ((0, above), (1, below)):
        s = 3 * p + other                                                              # colsum per input column
        if w == 1:
            out[v::2, 0] = (s[:, 0] * 4 + 8) >> 4
            out[v::2, 1] = (s[:, 0] * 4 + 7) >> 4
            continue
        left = np.concatenate([s[:, :1], s[:, :-1]], axis=1)
        right = np.concatenate([s[:, 1:], s[:, -1:]], axis=1)
        even = (3 * s + left + 8) >> 4
        odd = (3 * s + right + 7) >> 4
        even[:, 0] = (s[:, 0] * 4 + 8) >> 4
        odd[:, -1] = (s[:, -1] * 4 + 7) >> 4
        out[v::2, 0::2] = even
        out[v::2, 1::2] = odd
    return out.astype(np.uint8)


def _fix(x: float) -> int:
    return int(x * 65536 + 0.5)


def ycc_to_rgb(y: np.ndarray, cb: np.ndarray, cr: np.ndarray) -> np.ndarray:
    """jdcolor.c ycc_rgb_convert: 16-bit fixed-point tables, red / blue rounded per table entry, green per sum."""
    yy = y.astype(np.int32)
    b_, r_ = cb.astype(np.int32) - 128, cr.astype(np.int32) - 128
    cr_r = (_fix(1.40200) * r_ + 32768) >> 16
    cb_b = (_fix(1.77200) * b_ + 32768) >> 16
    g = ((-_fix(0.34414)) * b_ + 32768 + (-_fix(0.71414)) * r_) >> 16
    rgb = np.stack([yy + cr_r, yy + g, yy + cb_b], axis=-1)
    return np.clip(rgb, 0, 255).astype(np.uint8)


def decode(data: bytes, color: bool = True) -> np.ndarray:
    """cv2.imread semantics: BGR uint8 [H, W, 3] (color) or uint8 [H, W] (grey: the Y plane)."""
    info, coef = decode_coefficients(data)
    H, W = info["h"], info["w"]
    comps = info["comps"]
    planes = [idct_blocks(coef[i], info["qt"][c["tq"]]) for i, c in enumerate(comps)]
    if len(comps) == 1 or not color:
        yp = planes[0][:H, :W]
        return yp.copy() if not color else np.repeat(yp[:, :, None], 3, axis=2)
    if len(comps) != 3:
        raise JpegError("1 or 3 components expected")
    hmax, vmax = info["hmax"], info["vmax"]
    full = []
    for c, p in zip(comps, planes):
        dw, dh = -(-W * c["h"] // hmax), -(-H * c["v"] // vmax)                         # downsampled size: real rows / columns only
        p = p[:dh, :dw]
        if (c["h"], c["v"]) == (hmax, vmax):
            full.append(p)
        elif (hmax // c["h"], vmax // c["v"]) == (2, 1) and hmax % c["h"] == 0:
            # jdsample.c jinit_upsampler: the triangle filters are used only when the component is more than 2 samples wide,
            # narrower ones are replicated
            full.append(upsample_h2v1(p) if dw > 2 else np.repeat(p, 2, axis=1))
        elif (hmax // c["h"], vmax // c["v"]) == (2, 2) and hmax % c["h"] == 0 and vmax % c["v"] == 0:
            full.append(upsample_h2v2(p) if dw > 2 else np.repeat(np.repeat(p, 2, axis=0), 2, axis=1))
        else:
            raise JpegError("unsupported sampling factors %dx%d of %dx%d" % (c["h"], c["v"], hmax, vmax))
    rgb = ycc_to_rgb(full[0][:H, :W], full[1][:H, :W], full[2][:H, :W])
    return rgb[:, :, ::-1].copy()
