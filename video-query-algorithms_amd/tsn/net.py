"""Host side of the TSN forward pass: weights -> device blob, graph -> layer table, crops -> features.

``TsnNet`` is the handle over ``vq_tsn_*``.  It plays the role ``CaffeNet(net_proto, net_weights, gpu)``
plays in the reference (calcSig_wOF.py:52,55), but takes ALL B*T crops of a batch at once instead of one
snippet per call, and returns both the per-snippet ``global_pool`` blob (calcSig_wOF.py:95,112) and the
fp64 segment consensus (calcSig_wOF.py:82).
"""
from __future__ import annotations

import ctypes as C
import hashlib
import json
import os
from typing import Dict, Sequence

import numpy as np

from .. import _lib
from .._lib import ConvSegment, InputDesc, LayerDesc, TensorDesc, call
from .bn_inception import Graph, Plan

BK = 32                      # K granularity of the implicit-GEMM kernel (csrc/vq_tsn.hip)
RGB_MEAN = (104.0, 117.0, 123.0)      # BGR order, as cv2.imread delivers frames
FLOW_MEAN = (128.0,) * 10
_OPS = {"conv": _lib.VQ_OP_CONV, "maxpool": _lib.VQ_OP_MAXPOOL, "avgpool": _lib.VQ_OP_AVGPOOL,
        "gavgpool": _lib.VQ_OP_GLOBAL_AVGPOOL}


def synthetic_weights(graph: Graph, seed: int = 2) -> Dict[str, Dict[str, np.ndarray]]:
    """Random-init weights of the right architecture (SURVEY.md 8(d) cfg 2): conv ~ N(0, sqrt(2/fan_in)),
    small biases, frozen-BN statistics that keep activations O(1) through the 69 layers."""
    rng = np.random.default_rng(seed)
    w: Dict[str, Dict[str, np.ndarray]] = {}
    shapes = {graph.input_name: graph.input_shape[0]}
    for l in graph.layers:
        if l.type == "Convolution":
            cin = shapes[l.bottoms[0]]
            fan_in = cin * l.kernel * l.kernel
            w[l.name] = {
                "W": (rng.standard_normal((l.num_output, cin, l.kernel, l.kernel)) * np.sqrt(2.0 / fan_in)).astype(np.float32),
                "b": (rng.standard_normal(l.num_output) * 0.05).astype(np.float32)}
            shapes[l.tops[0]] = l.num_output
        elif l.type == "BN":
            c = shapes[l.bottoms[0]]
            w[l.name] = {"scale": rng.uniform(0.5, 1.5, c).astype(np.float32),
                         "shift": (rng.standard_normal(c) * 0.1).astype(np.float32),
                         "mean": (rng.standard_normal(c) * 0.1).astype(np.float32),
                         "var": rng.uniform(0.5, 1.5, c).astype(np.float32)}
            shapes[l.tops[0]] = c
        elif l.type == "Concat":
            shapes[l.tops[0]] = sum(shapes[b] for b in l.bottoms)
        elif l.type == "InnerProduct":
            shapes[l.tops[0]] = l.num_output
        else:
            shapes[l.tops[0]] = shapes[l.bottoms[0]]
    return w


def fold_bn(conv: Dict[str, np.ndarray], bn: Dict[str, np.ndarray] | None, eps: float = 1e-5, keep64: bool = False):
    """Frozen BN is a per-channel affine; fold it into the convolution: W' = a W, b' = a (b - mean) + shift."""
    W = conv["W"].astype(np.float64)
    b = conv["b"].astype(np.float64)
    if bn is not None:
        a = bn["scale"].astype(np.float64) / np.sqrt(bn["var"].astype(np.float64) + eps)
        W = W * a[:, None, None, None]
        b = a * (b - bn["mean"].astype(np.float64)) + bn["shift"].astype(np.float64)
    return (W if keep64 else W.astype(np.float32)), b.astype(np.float32)


# Winograd F(2x2, 3x3) filter transform (Lavin & Gray): U = G g G^T, 4x4 per (cout, cin)
_WINO_G = np.array([[1.0, 0.0, 0.0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0.0, 0.0, 1.0]])


def winograd_filters(W: np.ndarray) -> np.ndarray:
    """[Cout][Cin][3][3] (fp64, BN folded) -> device layout [Cin/8][16][Cout][8] fp32 (csrc/vq_wino.hip): transformed
    in fp64 and rounded once."""
    cout, cin = W.shape[:2]
    U = np.matmul(np.matmul(_WINO_G, W.astype(np.float64)), _WINO_G.T)                  # G g G^T per (o, c): [o][c][i][j] (G's entries are
                                                                                         # 0, 1/2, 1: every product is exact in fp64)
    U = U.reshape(cout, cin // 8, 8, 16).transpose(1, 3, 0, 2)                           # [c/8][xi][o][c%8]
    return np.ascontiguousarray(U, dtype=np.float32)


WINO16_PERM = (0, 2, 8, 10, 4, 6, 12, 14, 1, 3, 9, 11, 5, 7, 13, 15)     # csrc/vq_wino.hip: slot p of a 16-channel group holds channel PERM[p]
WINO16_MAX_MAP = 14                                                      # maps of at most 14 x 14 run the 16-tile units


def winograd_filters16(W: np.ndarray) -> np.ndarray:
    """The same transformed filters for the 16-tile units (VQ_OP_CONV_WINOGRAD16): [Cin/16][16][Cout][16 slots], slot p of a group =
    input channel WINO16_PERM[p] -- the order in which v_mfma_f32_16x16x4 must meet the channels to sum them as the 32-tile kernel does."""
    cout, cin = W.shape[:2]
    U = np.matmul(np.matmul(_WINO_G, W.astype(np.float64)), _WINO_G.T)                  # [o][c][i][j]
    U = U.reshape(cout, cin // 16, 16, 16)[:, :, list(WINO16_PERM), :].transpose(1, 3, 0, 2)   # [c/16][xi][o][slot]
    return np.ascontiguousarray(U, dtype=np.float32)


def wino16_default() -> bool:
    """Winograd layers on maps of at most 14 x 14 carry the filter layout of the 16-tile units too (the tiling table then picks 16 or 32
    tiles per launch) unless VQ_TSN_WINO16=0 (same bits either way)."""
    return os.environ.get("VQ_TSN_WINO16", "1") != "0"


def winograd_default() -> bool:
    """3x3 stride-1 convolutions run in Winograd form unless VQ_TSN_WINOGRAD=0."""
    return os.environ.get("VQ_TSN_WINOGRAD", "1") != "0"


def pad4(c: int) -> int:
    return (c + 3) // 4 * 4


def _round_up(x: int, m: int) -> int:
    return (x + m - 1) // m * m


def s2d_stem_weights(W: np.ndarray) -> np.ndarray:
    """[Cout][C][k][k] of a stride-2 convolution -> [Cout][k2][k2][4C] (k2 = ceil(k/2)) for the space-to-depth input
    slot0[Y][X][(p*2+q)*C + c]: W2[o][a][b][(p*2+q)*C + c] = W[o][c][2a+p][2b+q] (zero where 2a+p or 2b+q reaches k)."""
    cout, c, k, _ = W.shape
    k2 = (k + 1) // 2
    out = np.zeros((cout, k2, k2, 4 * c), dtype=W.dtype)
    for a in range(k2):
        for b in range(k2):
            for p in range(2):
                for q in range(2):
                    if 2 * a + p < k and 2 * b + q < k:
                        out[:, a, b, (p * 2 + q) * c:(p * 2 + q + 1) * c] = W[:, :, 2 * a + p, 2 * b + q]
    return out


def s2d_stem_weights_xmajor(W: np.ndarray) -> np.ndarray:
    """[Cout][c][7][7] of the stride-2 stem -> [Cout][K] for the x-major space-to-depth slot (vq_input_desc.s2d_order = 1:
    slot0[Y][X][(q*2+p)*c + ch] = crop[2Y+p-pad][2X+q-pad][ch]).  Kernel row r (cell rows r = 0..3) reads ONE run of the slot: x-tap
    t = 0..6, then (p, ch) -- 14 c floats, position 2 c t + c p + ch -- padded to whole 16-byte chunks against zero weights (RGB: 42 -> 44,
    flow stack: 140).  The four runs advance together, four floats per K-step: packed index 16 s + 4 r + e for run position 4 s + e;
    K = 176 (RGB) / 560 (flow).  Zero where 2r + p = 7."""
    cout, c, k, _ = W.shape
    if k != 7:
        raise ValueError("the x-major stem packing is for 7x7 kernels")
    steps = (14 * c + 3) // 4
    out = np.zeros((cout, steps, 4, 4), dtype=W.dtype)
    for r in range(4):
        for t in range(7):
            for p in range(2):
                if 2 * r + p < k:
                    for ch in range(c):
                        pos = 2 * c * t + c * p + ch
                        out[:, pos // 4, r, pos % 4] = W[:, ch, 2 * r + p, t]
    return out.reshape(cout, steps * 16)


_PACKER_DIGEST = None
_DEFAULT_TILES = None
DEFAULT_TILES_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "default_tiles.json")


def _default_tiles() -> dict:
    """{graph key: {"<n_crops>" | "<n_crops>p": [[BM, BN, BK, pipelined] per layer]}}: the tiling tables shipped with the library
    (measured on an MI355X by tools/make_default_tiles.py for both BN-Inception streams at the BASELINE batch sizes)."""
    global _DEFAULT_TILES
    if _DEFAULT_TILES is None:
        try:
            with open(DEFAULT_TILES_PATH) as f:
                _DEFAULT_TILES = json.load(f).get("tables", {})
        except (OSError, ValueError):
            _DEFAULT_TILES = {}
    return _DEFAULT_TILES




def packer_digest() -> str:
    """Names the CODE that turns weights into the packed blob and the layer table: this module (BN folding, Winograd filters, the
    space-to-depth stem, the LayerDesc fields it fills) and the plan builder (tsn/bn_inception.py: fusion, slots, channel offsets),
    plus the ctypes layout of the descriptors.  Part of the weight-cache key: a change to any of them makes every cached blob
    unreachable instead of silently wrong (a stale tiling table costs time; a stale blob would cost the features)."""
    global _PACKER_DIGEST
    if _PACKER_DIGEST is None:
        h = hashlib.sha1()
        here = os.path.dirname(os.path.abspath(__file__))
        for name in ("net.py", "bn_inception.py"):
            with open(os.path.join(here, name), "rb") as f:
                h.update(f.read())
        h.update(repr([(n, t.__name__) for n, t in LayerDesc._fields_] + [(n, t.__name__) for n, t in ConvSegment._fields_]).encode())
        _PACKER_DIGEST = h.hexdigest()
    return _PACKER_DIGEST


def _blob_hash(blob: np.ndarray, algo: str | None = None) -> str:
    """``<algo>:<digest>`` of the packed blob's bytes: xxh3 where the module is there (5 ms for the 42 MB of a network; the check runs on
    every cached build of an extractor), SHA-1 otherwise (37 ms).  A reader verifies with the algorithm the writer named."""
    raw = np.ascontiguousarray(blob).view(np.uint8)
    if algo in (None, "xxh3"):
        try:
            import xxhash
            return "xxh3:" + xxhash.xxh3_64_hexdigest(raw)
        except ImportError:
            if algo == "xxh3":
                raise
    return "sha1:" + hashlib.sha1(raw).hexdigest()


def _load_packed(path: str, n_ops: int):
    """(blob, layers, segments, conv_kp) of a packed network from the weight cache, or None -- also when the blob's bytes do not
    have the checksum stored with them (a torn or edited file)."""
    try:
        with open(path + ".json") as f:
            meta = json.load(f)
        if meta["n_ops"] != n_ops:
            return None
        blob = np.load(path + ".npy", mmap_mode="r", allow_pickle=False)
        if blob.dtype != np.float32 or blob.ndim != 1 or blob.size != meta["blob_floats"]:
            return None
        blob = np.ascontiguousarray(blob)
        if _blob_hash(blob, meta["blob_hash"].split(":", 1)[0]) != meta["blob_hash"]:
            return None
        layers = (LayerDesc * n_ops).from_buffer_copy(bytes.fromhex(meta["layers"]))
        n_seg = meta["n_segments"]
        segs = list((ConvSegment * n_seg).from_buffer_copy(bytes.fromhex(meta["segments"]))) if n_seg else []
        return np.ascontiguousarray(blob), layers, segs, {int(k): v for k, v in meta["conv_kp"].items()}
    except (OSError, ValueError, KeyError, ImportError):
        return None


def _store_packed(path: str, blob, layers, seg_list, conv_kp) -> None:
    """Best effort, atomic per file (several ranks of a fan-out may store the same network at once); the .json goes last and is
    what a reader looks for first."""
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        tmp = "%s.%d.tmp" % (path, os.getpid())
        with open(tmp + ".npy", "wb") as f:
            np.save(f, blob, allow_pickle=False)
        os.replace(tmp + ".npy", path + ".npy")
        segs = (ConvSegment * max(len(seg_list), 1))(*seg_list)
        meta = {"n_ops": len(layers), "blob_floats": int(blob.size), "blob_hash": _blob_hash(blob),
                "layers": bytes(layers).hex(), "n_segments": len(seg_list),
                "segments": bytes(segs).hex() if seg_list else "", "conv_kp": {str(k): int(v) for k, v in conv_kp.items()}}
        with open(tmp + ".json", "w") as f:
            json.dump(meta, f)
        os.replace(tmp + ".json", path + ".json")
    except OSError:
        pass


class TsnNet:
    def __init__(self, graph: Graph, weights, max_crops: int = 96, device: int = 0,
                 feature_blob: str = "global_pool", bn_eps: float = 1e-5, fuse: bool = True,
                 winograd: bool | None = None, stem_s2d: bool | None = None, cache_key: str | None = None,
                 tune_cache: str | None = None):
        """``weights``: {layer: {field: array}}, or a function without arguments that returns it (called only when the packed form
        is not in the cache).  ``cache_key``: names the weights' CONTENT (a digest of the file they come from, "synthetic:<seed>");
        with it the packed device blob -- BN folded, GEMM / Winograd layouts, biases -- is kept next to the library
        (``VQ_WEIGHT_CACHE=<dir>`` moves it, ``=0`` disables) under a digest of (key, layer plan, packing options, library ABI), and
        a later handle of the same network skips reading, folding and transforming the weights (0.3-0.5 s of every command-line run)."""
        self.graph = graph
        self.plan: Plan = graph.plan(feature_blob, fuse=fuse)
        self.in_channels = graph.input_shape[0]
        self.max_crops = int(max_crops)
        self.device = device
        self.winograd = winograd_default() if winograd is None else bool(winograd)
        self.wino16 = wino16_default()
        plan = self.plan
        cin_pad = pad4(self.in_channels)
        self.in_h, self.in_w = plan.tensors[0].h, plan.tensors[0].w
        # Space-to-depth stem: a k x k / stride-2 first convolution on very few channels is evaluated as a
        # ceil(k/2)^2 / stride-1 convolution over the 2x2-decimated input with 4x the channels -- K has no channel padding
        # (7x7x3: 16 taps x 12 = 192 instead of 49 x 4 -> 224), every load is 16-byte aligned and, being padding-free,
        # the layer runs on the aligned kernel path with its kernel rows folded into the channel axis.  Taken when the
        # packed K does not grow by more than 6 % (RGB: 224 -> 192; the 10-channel flow stack: 608 -> 640, still 7 %
        # faster because the per-chunk tap decoding disappears).
        stem = [op for op in plan.ops if op.src == 0]
        self.stem_s2d = False
        mode = "1" if stem_s2d is None else ("2" if stem_s2d else "0")   # None: when K does not grow; True: whenever possible; False: never
        if mode != "0" and len(stem) == 1:
            op = stem[0]
            k2 = (op.k + 1) // 2
            if op.kind == "conv" and op.stride == 2 and not op.segments and op.k >= 3 and \
                    (mode == "2" or _round_up(k2 * k2 * 4 * self.in_channels, BK) <= 1.06 * _round_up(op.k * op.k * cin_pad, BK)):
                self.stem_s2d = True
        # x-major cells + row-interleaved packing for the RGB 7x7 stem (K = 176 instead of 192: conv1 0.250 -> 0.238 ms at 96 crops).
        # The same form exists for any channel count (flow stack: K = 560 instead of 640) but LOSES there: 3.44 against 3.31 ms at 448
        # crops -- the flow stem is bound by cache traffic (a 40-float cell is read by 16 (output pixel, kernel row) pairs), and four 16-byte
        # chunks from four rows per lane group coalesce worse than 64 contiguous bytes.  So: x-major for <= 4 channels, (p,q,c) otherwise.
        self.stem_xmajor = bool(self.stem_s2d and stem[0].k == 7 and self.in_channels <= 4)
        in_slot_c = 4 * self.in_channels if self.stem_s2d else cin_pad
        tensors = (TensorDesc * len(plan.tensors))()
        for i, t in enumerate(plan.tensors):
            tensors[i] = TensorDesc(t.h, t.w, t.c)
        if self.stem_s2d:
            to_ = plan.tensors[stem[0].dst]
            tensors[0] = TensorDesc(to_.h + (stem[0].k + 1) // 2 - 1, to_.w + (stem[0].k + 1) // 2 - 1, in_slot_c)
        else:
            tensors[0] = TensorDesc(self.in_h, self.in_w, cin_pad)
        self.conv_kp = {}                                  # op index -> packed K of the direct kernel (bench accounting)
        cache_file = None
        where = os.environ.get("VQ_WEIGHT_CACHE", os.path.join(os.path.dirname(_lib.LIB_PATH), ".weight_cache"))
        if cache_key is not None and where != "0":
            opts = [_lib.ABI_VERSION, packer_digest(), BK, cache_key, repr(plan.ops), repr(plan.tensors), bool(self.winograd), bool(self.wino16), bool(self.stem_s2d), bool(self.stem_xmajor), float(bn_eps), cin_pad]
            cache_file = os.path.join(where, hashlib.sha1(json.dumps(opts).encode()).hexdigest()[:24])
        cached = _load_packed(cache_file, len(plan.ops)) if cache_file else None
        if cached is not None:
            blob, layers, seg_list, self.conv_kp = cached
        else:
            if callable(weights):
                weights = weights()
            layers = (LayerDesc * len(plan.ops))()
            seg_list = []
            chunks = []
            off = 0

            def packed_conv(name, bn, cin, cout, k, cin_dev, with_bias=True):
                W, b = fold_bn(weights[name], weights[bn] if bn else None, bn_eps)
                if W.shape != (cout, cin, k, k):
                    raise ValueError("weights of %s have shape %s, expected %s" % (name, W.shape, (cout, cin, k, k)))
                ohwi = np.zeros((cout, k, k, cin_dev), dtype=np.float32)
                ohwi[..., :cin] = W.transpose(0, 2, 3, 1)                      # [Cout][kh][kw][Cin]
                kdim = k * k * cin_dev
                kp = (kdim + BK - 1) // BK * BK
                packed = np.zeros((cout, kp), dtype=np.float32)
                packed[:, :kdim] = ohwi.reshape(cout, kdim)
                return packed, (b if with_bias else np.zeros_like(b))

            for i, op in enumerate(plan.ops):
                d = LayerDesc(op=_OPS[op.kind], src=op.src, dst=op.dst, src_coff=op.src_coff, dst_coff=op.dst_coff,
                              cin=op.cin, cout=op.cout, k=op.k, stride=op.stride, pad=op.pad, relu=int(op.relu),
                              ceil_mode=1, has_bias=0, seg_first=0, seg_count=0, w_off=0, b_off=0,
                              pre_pool_k=op.pre_pool[0] if op.pre_pool else 0, pre_pool_stride=op.pre_pool[1] if op.pre_pool else 0)
                if op.kind == "conv":
                    cin_dev = cin_pad if op.src == 0 else op.cin
                    if op.segments:                                            # sibling 1x1 convolutions: one GEMM
                        parts = [packed_conv(sg.name, sg.bn, op.cin, sg.cout, 1, cin_dev, sg.bias) for sg in op.segments]
                        packed = np.concatenate([q[0] for q in parts], axis=0)
                        b = np.concatenate([q[1] for q in parts])
                        d.seg_first, d.seg_count = len(seg_list), len(op.segments)
                        for sg in op.segments:
                            seg_list.append(ConvSegment(sg.cout, sg.dst, sg.dst_coff, int(sg.relu)))
                    elif (self.winograd and op.k == 3 and op.stride == 1 and op.pad == 1
                          and op.cin % 8 == 0 and op.cout % 32 == 0):
                        W64, b = fold_bn(weights[op.name], weights[op.bn] if op.bn else None, bn_eps, keep64=True)
                        if W64.shape != (op.cout, op.cin, 3, 3):
                            raise ValueError("weights of %s have shape %s" % (op.name, W64.shape))
                        src_t = plan.tensors[op.src]
                        if self.wino16 and op.cin % 16 == 0 and max(src_t.h, src_t.w) <= WINO16_MAX_MAP:
                            # few tiles per launch: both layouts, the tiling table picks units of 32 or of 16 tiles per launch (same bits)
                            packed = np.concatenate([winograd_filters(W64).reshape(-1), winograd_filters16(W64).reshape(-1)])
                            d.op = _lib.VQ_OP_CONV_WINOGRAD16
                        else:
                            packed = winograd_filters(W64)
                            d.op = _lib.VQ_OP_CONV_WINOGRAD
                        if not op.bias:
                            b = np.zeros_like(b)
                    elif self.stem_s2d and op.src == 0:
                        W, b = fold_bn(weights[op.name], weights[op.bn] if op.bn else None, bn_eps)
                        if W.shape != (op.cout, op.cin, op.k, op.k):
                            raise ValueError("weights of %s have shape %s" % (op.name, W.shape))
                        w2 = s2d_stem_weights(W)
                        cin_dev = in_slot_c
                        kdim = w2.shape[1] * w2.shape[2] * cin_dev
                        if self.stem_xmajor:
                            packed = np.ascontiguousarray(s2d_stem_weights_xmajor(W), dtype=np.float32)
                        else:
                            packed = np.zeros((op.cout, _round_up(kdim, BK)), dtype=np.float32)
                            packed[:, :kdim] = w2.reshape(op.cout, kdim)
                        if not op.bias:
                            b = np.zeros_like(b)
                        d.k, d.stride, d.pad = w2.shape[1], 1, 0
                    else:
                        packed, b = packed_conv(op.name, op.bn, op.cin, op.cout, op.k, cin_dev, op.bias)
                    if d.op == _lib.VQ_OP_CONV:
                        self.conv_kp[i] = packed.shape[1]
                    d.cin = cin_dev
                    d.w_off = off
                    chunks.append(packed.reshape(-1))
                    off += packed.size
                    d.b_off = off
                    bias = np.zeros(pad4(op.cout), dtype=np.float32)
                    bias[:op.cout] = b
                    chunks.append(bias)
                    off += bias.size
                elif op.kind == "avgpool" and op.bias_from is not None:       # finishes a commuted projection
                    _, b = fold_bn(weights[op.bias_from[0]], weights[op.bias_from[1]] if op.bias_from[1] else None, bn_eps)
                    d.has_bias = 1
                    d.b_off = off
                    bias = np.zeros(pad4(op.cout), dtype=np.float32)
                    bias[:op.cout] = b
                    chunks.append(bias)
                    off += bias.size
                elif op.src == 0:
                    d.cin = d.cout = cin_pad
                layers[i] = d
            blob = np.ascontiguousarray(np.concatenate(chunks))
            if cache_file:
                _store_packed(cache_file, blob, layers, seg_list, self.conv_kp)
        segs = (ConvSegment * max(len(seg_list), 1))(*seg_list)
        self._h = C.c_void_p()
        inp = InputDesc(self.in_h, self.in_w, self.in_channels, stem[0].pad if self.stem_s2d else -1, stem[0].k if self.stem_s2d else 0,
                        1 if self.stem_xmajor else 0)
        call("vq_tsn_create", tensors, len(plan.tensors), layers, len(plan.ops), segs, len(seg_list),
             blob.ctypes.data_as(C.c_void_p), blob.size, C.byref(inp), plan.feature_slot, self.max_crops, device,
             C.byref(self._h))
        self.feature_dim = plan.feature_dim
        self._layer_ops = [int(layers[i].op) for i in range(len(plan.ops))]
        self._tensor_c = [tensors[i].c for i in range(len(plan.tensors))]
        self._tensor_hw = [(tensors[i].h, tensors[i].w) for i in range(len(plan.tensors))]
        # Tiling tables survive the process: the first forward of a batch size times 24 tilings per layer (~2 s of
        # launches); the winners are kept per (layer table, library ABI) next to the library and installed at creation.
        # Every tiling gives the same bits, so a stale or foreign table can only cost speed.  VQ_TUNE_CACHE=0 (or tune_cache="0")
        # disables, VQ_TUNE_CACHE=<dir> moves the files.
        desc = [(d.op, d.src, d.dst, d.src_coff, d.dst_coff, d.cin, d.cout, d.k, d.stride, d.pad, d.seg_count, d.pre_pool_k) for d in layers]
        shapes = [(tensors[i].h, tensors[i].w, tensors[i].c) for i in range(len(plan.tensors))]
        self._tune_file = None
        self._tune_saved = set()        # (n_crops, paired) of the tables that are in the cache file or came with the library
        self._tune_checked = set()      # batch sizes whose first forward (the one that tunes) is behind us
        where = tune_cache if tune_cache is not None else os.environ.get("VQ_TUNE_CACHE", os.path.join(os.path.dirname(_lib.LIB_PATH), ".tune_cache"))
        split = os.environ.get("VQ_TSN_SPLIT", "2")           # a paired table was timed beside this many twins
        self.graph_key = hashlib.sha1(json.dumps([desc, shapes]).encode()).hexdigest()[:20]

        def install(table):
            for name, tiles in table.items():
                if name == "one_stream":                      # batch sizes measured faster on one stream than as two sub-batches
                    for n_crops in tiles:
                        self.set_split(int(n_crops), False)
                    continue
                paired = name.endswith("p")
                n_crops = int(name[:-1] if paired else name)
                if n_crops <= self.max_crops and (n_crops, paired) not in self._tune_saved:
                    self._install_tiles(n_crops, np.array(tiles, dtype=np.int32), paired)
                    self._tune_saved.add((n_crops, paired))
        if where != "0":
            key = hashlib.sha1(json.dumps([_lib.ABI_VERSION, desc, shapes, split]).encode()).hexdigest()[:20]
            self._tune_file = os.path.join(where, key + ".json")
            try:
                with open(self._tune_file) as f:
                    install(json.load(f))
            except (OSError, ValueError, _lib.VqError):
                pass            # no file yet, or one written for another kernel set: tune afresh
        # The tables shipped with the library (tsn/default_tiles.json: the BASELINE shapes on gfx950, tools/make_default_tiles.py) fill in
        # what this machine has not measured itself, so that a cold process times nothing.  VQ_TSN_AUTOTUNE=1 leaves them out: every
        # size is then swept in its first forward (and kept in the cache above) -- the refinement for a machine that differs.
        self.default_tables = 0
        if os.environ.get("VQ_TSN_AUTOTUNE") != "1" and os.environ.get("VQ_TSN_DEFAULT_TILES", "1") != "0" and split == "2":
            try:
                before = len(self._tune_saved)
                install(_default_tiles().get(self.graph_key, {}))
                self.default_tables = len(self._tune_saved) - before
            except (ValueError, _lib.VqError):
                pass            # a table written for another kernel set

    def _persist_tuning(self, n_crops):
        if self._tune_file is None or n_crops in self._tune_checked:
            return
        self._tune_checked.add(n_crops)
        sizes = {(n, p) for n, p, borrowed in self.tile_tables() if not borrowed}      # measured here (or installed from a file)
        if sizes <= self._tune_saved:
            return
        try:
            os.makedirs(os.path.dirname(self._tune_file), exist_ok=True)
            table = {}
            try:
                with open(self._tune_file) as f:
                    table = json.load(f)
            except (OSError, ValueError):
                pass
            for n, p in sizes - self._tune_saved:
                table["%d%s" % (n, "p" if p else "")] = self.layer_tiles(n, paired=p).tolist()
            tmp = "%s.%d.tmp" % (self._tune_file, os.getpid())
            with open(tmp, "w") as f:
                json.dump(table, f)
            os.replace(tmp, self._tune_file)
            self._tune_saved |= sizes
        except OSError:
            self._tune_file = None      # read-only tree: keep tuning per process

    # ------------------------------------------------------------------
    def set_stream(self, hip_stream: int):
        call("vq_tsn_set_stream", self._h, C.c_void_p(hip_stream))

    def forward(self, crops: np.ndarray, T: int, mean: Sequence[float], want_per_snippet: bool = True):
        """crops uint8 [B*T, H, W, C] (the T snippets of a clip contiguous) ->
        (consensus [B, D] fp64, per-snippet global_pool [B*T, D] fp32)."""
        x = np.ascontiguousarray(crops, dtype=np.uint8)
        if x.ndim != 4 or x.shape[1:] != (self.in_h, self.in_w, self.in_channels):
            raise ValueError("crops must be [n,%d,%d,%d] uint8" % (self.in_h, self.in_w, self.in_channels))
        n = x.shape[0]
        if n % T:
            raise ValueError("number of crops must be a multiple of T")
        m = np.ascontiguousarray(mean, dtype=np.float32)
        if m.shape != (self.in_channels,):
            raise ValueError("mean needs one entry per input channel")
        feat = np.empty((n // T, self.feature_dim), dtype=np.float64)
        ps = np.empty((n, self.feature_dim), dtype=np.float32) if want_per_snippet else None
        call("vq_tsn_forward", self._h, x.ctypes.data_as(C.c_void_p), 0, n, T, m.ctypes.data_as(C.POINTER(C.c_float)),
             feat.ctypes.data_as(C.c_void_p), ps.ctypes.data_as(C.c_void_p) if ps is not None else None)
        self._persist_tuning(n)
        return feat, ps

    def forward_device(self, crops_dev_ptr: int, n: int, T: int, mean: Sequence[float], feat_out: np.ndarray | None = None):
        """Crops already in HBM (uint8 NHWC); results stay on the device (see feat_devptr).  With ``feat_out`` (float64 [n / T, D],
        C-contiguous) the consensus features are also copied to the host behind the forward and the call returns when they are there:
        it waits for the handle's OWN stream only -- not, as a device-wide synchronisation would, for the decode kernels of later batches
        that other threads have in flight on other streams."""
        m = np.ascontiguousarray(mean, dtype=np.float32)
        host = None
        if feat_out is not None:
            if feat_out.dtype != np.float64 or not feat_out.flags.c_contiguous or feat_out.shape != (n // T, self.feature_dim):
                raise ValueError("feat_out must be a C-contiguous float64 array of shape (%d, %d)" % (n // T, self.feature_dim))
            host = feat_out.ctypes.data_as(C.c_void_p)
        call("vq_tsn_forward", self._h, C.c_void_p(crops_dev_ptr), 1, n, T, m.ctypes.data_as(C.POINTER(C.c_float)), host, None)
        self._persist_tuning(n)
        return feat_out

    def features_tensor(self, n_clips: int):
        """Zero-copy torch view [n_clips, D] fp64 of the consensus features of the last forward (device memory owned
        by the handle: clone it before the next forward).  Synchronises the device first."""
        _lib.require_torch_runtime("TsnNet.features_tensor")
        import torch
        f, _ = self.feat_devptr()

        class _V:
            __cuda_array_interface__ = {"shape": (int(n_clips), self.feature_dim), "typestr": "<f8", "data": (int(f), False), "version": 2}
        dev = torch.device("cuda", self.device)
        torch.cuda.synchronize(dev)
        return torch.as_tensor(_V(), device=dev)

    def read_features(self, out: np.ndarray):
        """Copy the consensus features [B, D] fp64 of the last forward_device to the host (synchronises)."""
        out[...] = self.features_tensor(out.shape[0]).cpu().numpy()
        return out

    def feat_devptr(self):
        f, p = C.c_void_p(), C.c_void_p()
        call("vq_tsn_feat_devptr", self._h, C.byref(f), C.byref(p))
        return f.value, p.value

    def read_blob(self, name: str, n_crops: int) -> np.ndarray:
        """Activation of blob `name` from the last forward, NHWC [n_crops, h, w, c] (per-layer parity)."""
        slot, coff, c = self.plan.blob_loc[name]
        h, w = self._tensor_hw[slot]
        cs = self._tensor_c[slot]
        buf = np.empty((n_crops, h, w, cs), dtype=np.float32)
        call("vq_tsn_read_tensor", self._h, slot, n_crops, buf.ctypes.data_as(C.c_void_p))
        if slot == 0 and self.stem_s2d:                   # undo the space-to-depth packing of the input slot
            pad = [op for op in self.plan.ops if op.src == 0][0].pad
            ci = self.in_channels
            img = np.zeros((n_crops, 2 * h, 2 * w, ci), dtype=np.float32)
            for p in range(2):
                for q in range(2):
                    cell = (q * 2 + p) if self.stem_xmajor else (p * 2 + q)
                    img[:, p::2, q::2] = buf[..., cell * ci:(cell + 1) * ci]
            return img[:, pad:pad + self.in_h, pad:pad + self.in_w]
        return buf[..., coff:coff + c]

    def set_profile(self, depth: int, every: int = 1, split_between: bool = False):
        """depth > 0: keep HIP-event timings of the last `depth` profiled forwards (averaged by layer_times); 0 = off.
        every = n: only every n-th forward carries the events.  The forwards in between run the same launches on the same one
        stream, or -- ``split_between`` -- as the product runs them (sub-batches on separate streams): only the sampled ones
        then stay on one stream, where a launch's duration is the kernel alone."""
        call("vq_tsn_set_profile", self._h, int(depth))
        call("vq_tsn_set_profile_every", self._h, int(every))
        call("vq_tsn_set_profile_split", self._h, 1 if split_between else 0)

    def layer_times(self):
        """(names, kinds, mean ms[n_layers] over the profiled forwards, flops[n_layers] of the last batch)."""
        n = len(self.plan.ops)
        ms = np.empty(n, dtype=np.float32)
        fl = np.empty(n, dtype=np.float64)
        call("vq_tsn_layer_times", self._h, ms.ctypes.data_as(C.c_void_p), fl.ctypes.data_as(C.c_void_p), n)
        names = [(o.pre_pool[2] + ">" + o.name) if o.pre_pool else o.name for o in self.plan.ops]
        return names, [o.kind for o in self.plan.ops], ms, fl

    def layer_tiles(self, n_crops: int, paired: bool | None = None) -> np.ndarray:
        """[n_layers, 4] (BM, BN, BK, pipelined) implicit-GEMM tiling per conv layer at this batch size.  ``paired``: the table of a
        SUB-BATCH of this size (timed side by side on the sub-batch streams) / of a one-stream forward; None = the one-stream table,
        else the paired one, else the heuristic."""
        n = len(self.plan.ops)
        out = np.zeros((n, 4), dtype=np.int32)
        if paired is None:
            call("vq_tsn_layer_tiles", self._h, int(n_crops), out.ctypes.data_as(C.c_void_p), n)
        else:
            call("vq_tsn_get_tiles", self._h, int(n_crops), 1 if paired else 0, out.ctypes.data_as(C.c_void_p), n)
        return out

    def set_split(self, n_crops: int, split: bool | None):
        """Forwards of (about) n_crops crops run as sub-batches on separate streams (True, the default), on one stream (False: measured
        faster at that size), or as the default says again (None)."""
        call("vq_tsn_set_split", self._h, int(n_crops), -1 if split is None else (1 if split else 0))

    def tune(self, n_crops: int, paired: bool):
        """Run the timing sweep for one table now (the slots hold whatever they hold: only durations matter)."""
        call("vq_tsn_tune", self._h, int(n_crops), 1 if paired else 0)

    def tile_tables(self):
        """[(n_crops, paired, borrowed)] of every tiling table the handle holds (include/vq_amd.h: vq_tsn_tile_tables)."""
        k = C.c_int32()
        call("vq_tsn_tile_tables", self._h, None, None, 0, C.byref(k))
        sizes = np.zeros(max(k.value, 1), dtype=np.int32)
        flags = np.zeros(max(k.value, 1), dtype=np.int32)
        call("vq_tsn_tile_tables", self._h, sizes.ctypes.data_as(C.c_void_p), flags.ctypes.data_as(C.c_void_p), sizes.size, C.byref(k))
        return [(int(s), bool(f & 1), bool(f & 2)) for s, f in zip(sizes[:k.value], flags[:k.value])]

    def layer_op(self, i: int) -> int:
        """The op code layer i was lowered to (include/vq_amd.h: VQ_OP_*)."""
        return int(self._layer_ops[i])

    def launch_items(self):
        """(item_of_layer [n_layers], n_items): which kernel launch of a forward executes each layer -- the Winograd
        convolutions of one dependency level share a launch."""
        n = len(self.plan.ops)
        out = np.zeros(n, dtype=np.int32)
        k = C.c_int32()
        call("vq_tsn_launch_items", self._h, out.ctypes.data_as(C.c_void_p), n, C.byref(k))
        return out, k.value

    def tuned_sizes(self):
        """Batch sizes for which a tiling table exists (autotuned on first use, or installed)."""
        k = C.c_int32()
        call("vq_tsn_tuned_sizes", self._h, None, 0, C.byref(k))
        out = np.zeros(max(k.value, 1), dtype=np.int32)
        call("vq_tsn_tuned_sizes", self._h, out.ctypes.data_as(C.c_void_p), out.size, C.byref(k))
        return out[:k.value].tolist()

    def set_layer_tiles(self, n_crops: int, tiles: np.ndarray, paired: bool | None = None):
        """Install a tiling table (from layer_tiles, e.g. of an earlier process) instead of autotuning.  ``paired`` None: for a forward of
        n_crops however it runs (the one-stream table of that size AND the paired tables of the sub-batches the default forward cuts it
        into); True / False: that one table.  A handle whose tables were set by hand no longer writes the tiling cache: what it would
        save is the caller's choice (a test forcing one kernel, an A/B), not a measurement, and later sizes may borrow from it."""
        self._tune_file = None
        if paired is None:
            t = np.ascontiguousarray(tiles, dtype=np.int32)
            call("vq_tsn_set_layer_tiles", self._h, int(n_crops), t.ctypes.data_as(C.c_void_p), t.shape[0])
        else:
            self._install_tiles(n_crops, tiles, paired)

    def _install_tiles(self, n_crops: int, tiles: np.ndarray, paired: bool = False):
        t = np.ascontiguousarray(tiles, dtype=np.int32)
        call("vq_tsn_set_tiles", self._h, int(n_crops), 1 if paired else 0, t.ctypes.data_as(C.c_void_p), t.shape[0])

    def flops_per_crop(self) -> float:
        out = C.c_double()
        call("vq_tsn_flops_per_crop", self._h, C.byref(out))
        return out.value

    def close(self):
        if self._h:
            _lib.load().vq_tsn_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
