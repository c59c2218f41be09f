"""CPU oracle for hot path A (TSN BN-Inception forward + segment consensus).

TEST INFRASTRUCTURE ONLY -- never imported by the product package.

Parity status: **PARITY UNPINNED** for the network arithmetic.  The reference runs this path inside
third-party code that is not in its tree (yjxiong "caffe-action" fork + ``pyActionRecog``; no commit is
pinned anywhere, the PARC-fine-tuned ``.caffemodel`` weights and the source frames are not shipped;
SURVEY.md 8(c)).  What this file restates is therefore

* the network topology and hyper-parameters of
  ``src/features_GPU_compute/models/ucf101/tsn_bn_inception_{rgb,flow}_deploy.prototxt`` (walked layer
  by layer, NOT through the product's lowering),
* the published Caffe layer semantics (BVLC caffe ``conv_layer.cpp`` / ``pooling_layer.cpp``: floor-mode
  convolution size, ceil-mode pooling size with the last-window clip, average pooling dividing by the
  window clipped to the padded extent, max pooling ignoring padding; frozen BN = per-channel affine),
* the driver logic that IS in the reference tree: snippet ticks (calcSig_wOF.py:67-72, Python-2 integer
  division), flow-stack indices (:104), crop-0 feature (:95,112), consensus ``np.array(...).mean(axis=0)``
  (:82) -- these parts are pinned by the reference's own source text.

The forward is evaluated with torch CPU ops in float64 (or float32); ``conv_direct`` / ``pool_direct``
are loop-level numpy restatements used to pin the torch calls on small shapes.
"""
from __future__ import annotations

import math
from typing import Dict, List, Sequence

import numpy as np


# ------------------------------------------------------------------------------------------------
# driver logic restated from calcSig_wOF.py
# ------------------------------------------------------------------------------------------------
def frame_ticks(frame_cnt: int, num_frame_per_video: int, stack_depth: int) -> List[int]:
    """calcSig_wOF.py:67-72 (Python 2: ``/`` on ints floors; T == 1 divides by zero like the original)."""
    step = (frame_cnt - stack_depth) // (num_frame_per_video - 1)
    if step > 0:
        ticks = list(range(1, min(2 + step * (num_frame_per_video - 1), frame_cnt + 1), step))
    else:
        ticks = [1] * num_frame_per_video
    assert len(ticks) == num_frame_per_video
    return ticks


def flow_stack_indices(tick: int, frame_cnt: int, stack_depth: int) -> List[int]:
    """calcSig_wOF.py:104."""
    return [min(frame_cnt, tick + offset) for offset in range(stack_depth)]


def consensus(per_snippet: np.ndarray, T: int) -> np.ndarray:
    """calcSig_wOF.py:82: fp64 mean over the T snippets of each clip ([B*T,D] fp32 -> [B,D] fp64).

    ``.tolist()`` at :95 turns the fp32 blob into Python floats, ``np.array(...).mean(axis=0)`` then
    reduces the (T,1,D) float64 array along axis 0: a sequential row sum divided by T.
    """
    ps = np.asarray(per_snippet)
    b = ps.shape[0] // T
    out = np.empty((b, ps.shape[1]), dtype=np.float64)
    for i in range(b):
        frame_features = [ps[i * T + j].reshape(1, -1).tolist() for j in range(T)]
        out[i] = np.array(frame_features).mean(axis=0)[0]
    return out


# ------------------------------------------------------------------------------------------------
# Caffe layer semantics, loop level (small shapes only)
# ------------------------------------------------------------------------------------------------
def conv_out(size, k, s, p):
    return (size + 2 * p - k) // s + 1


def pool_out(size, k, s, p):
    out = int(math.ceil((size + 2 * p - k) / float(s))) + 1
    if p > 0 and (out - 1) * s >= size + p:
        out -= 1
    return out


def conv_direct(x: np.ndarray, w: np.ndarray, b: np.ndarray, stride: int, pad: int) -> np.ndarray:
    """Cross-correlation, NCHW, weights [Cout][Cin][kh][kw] (Caffe ConvolutionLayer)."""
    n, c, h, wd = x.shape
    co, ci, kh, kw = w.shape
    ho, wo = conv_out(h, kh, stride, pad), conv_out(wd, kw, stride, pad)
    xp = np.zeros((n, c, h + 2 * pad, wd + 2 * pad), dtype=np.float64)
    xp[:, :, pad:pad + h, pad:pad + wd] = x
    out = np.zeros((n, co, ho, wo), dtype=np.float64)
    for oh in range(ho):
        for ow in range(wo):
            patch = xp[:, :, oh * stride:oh * stride + kh, ow * stride:ow * stride + kw]
            out[:, :, oh, ow] = np.tensordot(patch, w.astype(np.float64), axes=([1, 2, 3], [1, 2, 3]))
    return out + b.astype(np.float64)[None, :, None, None]


def pool_direct(x: np.ndarray, k: int, stride: int, pad: int, mode: str) -> np.ndarray:
    """Caffe PoolingLayer::Forward_cpu for MAX and AVE."""
    n, c, h, w = x.shape
    ho, wo = pool_out(h, k, stride, pad), pool_out(w, k, stride, pad)
    out = np.empty((n, c, ho, wo), dtype=x.dtype)
    for ph in range(ho):
        for pw in range(wo):
            hs, ws = ph * stride - pad, pw * stride - pad
            he, we = min(hs + k, h + pad), min(ws + k, w + pad)
            pool_size = (he - hs) * (we - ws)
            hs, ws, he, we = max(hs, 0), max(ws, 0), min(he, h), min(we, w)
            win = x[:, :, hs:he, ws:we]
            if mode == "MAX":
                out[:, :, ph, pw] = win.max(axis=(2, 3))
            else:
                acc = np.zeros((n, c), dtype=x.dtype)
                for i in range(hs, he):                    # sequential accumulation, h then w
                    for j in range(ws, we):
                        acc = acc + x[:, :, i, j]
                out[:, :, ph, pw] = acc / x.dtype.type(pool_size)
    return out


# ------------------------------------------------------------------------------------------------
# forward pass over the layer list (graph = product's parsed prototxt, passed in as plain data)
# ------------------------------------------------------------------------------------------------
def bn_affine(bn: Dict[str, np.ndarray], eps: float = 1e-5):
    """Frozen BN as a per-channel affine: y = a*x + c, a = scale/sqrt(var+eps), c = shift - a*mean."""
    a = bn["scale"].astype(np.float64) / np.sqrt(bn["var"].astype(np.float64) + eps)
    c = bn["shift"].astype(np.float64) - a * bn["mean"].astype(np.float64)
    return a, c


def forward(layers: Sequence, input_name: str, weights: Dict[str, Dict[str, np.ndarray]], x_nchw: np.ndarray,
            dtype=np.float64, keep: Sequence[str] = (), threads: int | None = None) -> Dict[str, np.ndarray]:
    """Evaluate the layer list.  ``layers`` items need attributes name/type/bottoms/tops/num_output/kernel/
    stride/pad/pool.  Returns {blob: NCHW array} for ``keep`` (all blobs if keep is None)."""
    import torch
    import torch.nn.functional as F
    if threads:
        torch.set_num_threads(threads)
    td = torch.float64 if np.dtype(dtype) == np.float64 else torch.float32
    blobs = {input_name: torch.from_numpy(np.ascontiguousarray(x_nchw)).to(td)}
    out = {}
    last_use = {}
    for i, l in enumerate(layers):
        for b in l.bottoms:
            last_use[b] = i
    for i, l in enumerate(layers):
        if l.type == "Convolution":
            p = weights[l.name]
            y = F.conv2d(blobs[l.bottoms[0]], torch.from_numpy(p["W"]).to(td), torch.from_numpy(p["b"]).to(td),
                         stride=l.stride, padding=l.pad)
        elif l.type == "BN":
            a, c = bn_affine(weights[l.name])
            y = blobs[l.bottoms[0]] * torch.from_numpy(a).to(td)[None, :, None, None] \
                + torch.from_numpy(c).to(td)[None, :, None, None]
        elif l.type == "ReLU":
            y = torch.clamp_min(blobs[l.bottoms[0]], 0)
        elif l.type == "Pooling":
            xin = blobs[l.bottoms[0]]
            ho = pool_out(xin.shape[2], l.kernel, l.stride, l.pad)
            if l.pool == "MAX":
                y = F.max_pool2d(xin, l.kernel, l.stride, l.pad, ceil_mode=True)
            else:
                y = F.avg_pool2d(xin, l.kernel, l.stride, l.pad, ceil_mode=True, count_include_pad=True)
            assert y.shape[2] == ho, (l.name, y.shape, ho)
        elif l.type == "Concat":
            y = torch.cat([blobs[b] for b in l.bottoms], dim=1)
        elif l.type == "Dropout":
            y = blobs[l.bottoms[0]]                       # caffe.TEST
        elif l.type == "InnerProduct":
            continue                                      # fc-action is not part of the feature path
        else:
            raise ValueError(l.type)
        blobs[l.tops[0]] = y
        if keep is None or l.tops[0] in keep:
            out[l.tops[0]] = y.numpy().copy() if keep is None else y.numpy()
        for b in l.bottoms:                               # free what is no longer needed
            if last_use.get(b) == i and b != l.tops[0] and (keep is not None and b not in keep):
                blobs.pop(b, None)
    return out


def preprocess(crops_u8_nhwc: np.ndarray, mean: Sequence[float]) -> np.ndarray:
    """uint8 NHWC crops -> float NCHW minus per-channel mean (Appendix B: BGR [104,117,123] / flow 128)."""
    x = crops_u8_nhwc.astype(np.float64) - np.asarray(mean, dtype=np.float64)[None, None, None, :]
    return np.ascontiguousarray(x.transpose(0, 3, 1, 2))


def features(layers, input_name, weights, crops_u8_nhwc, mean, T, dtype=np.float64, blob="global_pool",
             threads=None):
    """Crops [B*T,H,W,C] u8 -> (per_snippet [B*T,D] in `dtype`, consensus [B,D] fp64 over fp32-rounded snippets)."""
    x = preprocess(crops_u8_nhwc, mean)
    y = forward(layers, input_name, weights, x, dtype=dtype, keep=(blob,), threads=threads)[blob]
    ps = y.reshape(y.shape[0], -1)
    return ps, consensus(ps.astype(np.float32), T)
