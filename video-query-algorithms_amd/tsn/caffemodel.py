"""Reader (and, for tests, writer) of Caffe ``.caffemodel`` weight files without a protobuf schema or Caffe.

The reference passes ``.caffemodel`` paths to ``CaffeNet`` (calcSig_wOF.py:52,55; calcSig_wOF_ensemble.sh:15-17).
A ``.caffemodel`` is a serialized ``NetParameter``; only a handful of fields matter for a frozen forward pass, so the
protobuf wire format is decoded directly:

    NetParameter    : 1 name (string) | 100 layer (LayerParameter, repeated) | 2 layers (V1LayerParameter, repeated)
    LayerParameter  : 1 name | 2 type (string) | 7 blobs (BlobProto, repeated)
    V1LayerParameter: 4 name | 5 type (enum)   | 6 blobs
    BlobProto       : 7 shape (BlobShape: 1 dim, packed int64) | 5 data (packed float) | 8 double_data (packed double)
                      | 1 num, 2 channels, 3 height, 4 width (legacy 4-d shape)

PARITY UNPINNED: no ``.caffemodel`` ships with the reference (``.gitignore`` excludes them), so the meaning given to the
four blobs of the yjxiong-fork ``BN`` layer -- scale, shift, mean, variance, in that order -- and its epsilon default
follow SURVEY.md Appendix B ("from memory") and are parameters of :func:`weights_from_caffemodel`.
"""
from __future__ import annotations

import struct
from typing import Dict, List, Tuple

import numpy as np


# ------------------------------------------------------------------------------------------------ wire format
def _varint(buf: bytes, pos: int) -> Tuple[int, int]:
    result = shift = 0
    while True:
        b = buf[pos]
        pos += 1
        result |= (b & 0x7F) << shift
        if not b & 0x80:
            return result, pos
        shift += 7


def _fields(buf: bytes):
    """Yield (field number, wire type, value) of one message; length-delimited values are memoryview slices."""
    pos, n = 0, len(buf)
    mv = memoryview(buf)
    while pos < n:
        key, pos = _varint(buf, pos)
        num, wt = key >> 3, key & 7
        if wt == 0:
            val, pos = _varint(buf, pos)
        elif wt == 1:
            val = bytes(mv[pos:pos + 8])
            pos += 8
        elif wt == 2:
            ln, pos = _varint(buf, pos)
            val = mv[pos:pos + ln]
            pos += ln
        elif wt == 5:
            val = bytes(mv[pos:pos + 4])
            pos += 4
        else:
            raise ValueError("unsupported protobuf wire type %d" % wt)
        yield num, wt, val


def _blob(buf) -> np.ndarray:
    buf = bytes(buf)
    shape: List[int] = []
    legacy = {}
    chunks: List[np.ndarray] = []        # a repeated field may arrive packed, element by element, or as several packed runs
    for num, wt, val in _fields(buf):
        if num == 7 and wt == 2:                                   # BlobShape
            for n2, w2, v2 in _fields(bytes(val)):
                if n2 == 1 and w2 == 2:                            # packed dims
                    b2, p = bytes(v2), 0
                    while p < len(b2):
                        d, p = _varint(b2, p)
                        shape.append(d)
                elif n2 == 1 and w2 == 0:
                    shape.append(v2)
        elif num == 5 and wt in (2, 5):                            # data: packed run, or one fixed32 element
            chunks.append(np.frombuffer(bytes(val), dtype="<f4"))
        elif num == 8 and wt in (2, 1):                            # double_data
            chunks.append(np.frombuffer(bytes(val), dtype="<f8").astype(np.float32))
        elif num in (1, 2, 3, 4) and wt == 0:
            legacy[num] = val
    data = np.concatenate(chunks) if chunks else np.zeros(0, dtype=np.float32)
    if not shape and legacy:
        shape = [legacy.get(i, 1) for i in (1, 2, 3, 4)]
    return np.array(data, dtype=np.float32).reshape(shape) if shape else np.array(data, dtype=np.float32)


def read_caffemodel(path: str) -> Dict[str, Dict]:
    """{layer name: {"type": str | int, "blobs": [ndarray, ...]}} for every layer that carries blobs."""
    with open(path, "rb") as f:
        buf = f.read()
    out: Dict[str, Dict] = {}
    for num, wt, val in _fields(buf):
        if wt != 2 or num not in (100, 2):
            continue
        name_field, type_field, blob_field = (1, 2, 7) if num == 100 else (4, 5, 6)
        name, ltype, blobs = None, None, []
        for n2, w2, v2 in _fields(bytes(val)):
            if n2 == name_field and w2 == 2:
                name = bytes(v2).decode("utf-8")
            elif n2 == type_field:
                ltype = bytes(v2).decode("utf-8") if w2 == 2 else int(v2)
            elif n2 == blob_field and w2 == 2:
                blobs.append(_blob(v2))
        if name is not None and blobs:
            out[name] = {"type": ltype, "blobs": blobs}
    return out


def weights_from_caffemodel(path: str, graph, bn_blob_order=("scale", "shift", "mean", "var")) -> Dict[str, Dict[str, np.ndarray]]:
    """Weights dict in the layout ``TsnNet`` takes, for the layers of ``graph`` (a parsed deploy prototxt)."""
    layers = read_caffemodel(path)
    w: Dict[str, Dict[str, np.ndarray]] = {}
    for l in graph.layers:
        if l.type == "Convolution":
            if l.name not in layers:
                raise KeyError("caffemodel has no blobs for convolution %r" % l.name)
            blobs = layers[l.name]["blobs"]
            W = blobs[0]
            if W.ndim != 4:
                raise ValueError("convolution %s: weight blob has shape %s" % (l.name, W.shape))
            b = blobs[1].reshape(-1) if len(blobs) > 1 else np.zeros(W.shape[0], dtype=np.float32)
            w[l.name] = {"W": np.ascontiguousarray(W), "b": np.ascontiguousarray(b)}
        elif l.type == "BN":
            if l.name not in layers:
                raise KeyError("caffemodel has no blobs for BN layer %r" % l.name)
            blobs = layers[l.name]["blobs"]
            if len(blobs) != 4:
                raise ValueError("BN layer %s: expected 4 blobs (yjxiong caffe fork), got %d" % (l.name, len(blobs)))
            w[l.name] = {k: np.ascontiguousarray(blobs[i].reshape(-1)) for i, k in enumerate(bn_blob_order)}
    return w


# ------------------------------------------------------------------------------------------------ writer (tests)
def _enc_varint(v: int) -> bytes:
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        out.append(b | (0x80 if v else 0))
        if not v:
            return bytes(out)


def _enc_ld(num: int, payload: bytes) -> bytes:
    return _enc_varint((num << 3) | 2) + _enc_varint(len(payload)) + payload


def _enc_blob(a: np.ndarray) -> bytes:
    a = np.ascontiguousarray(a, dtype="<f4")
    shape = b"".join(_enc_varint(int(d)) for d in a.shape)
    return _enc_ld(7, _enc_ld(1, shape)) + _enc_ld(5, a.tobytes())


def write_caffemodel(path: str, graph, weights: Dict[str, Dict[str, np.ndarray]], name: str = "BN-Inception"):
    """Serialise weights as a NetParameter with the new-style ``layer`` field (what the TSN tooling saves)."""
    out = bytearray(_enc_ld(1, name.encode()))
    for l in graph.layers:
        if l.name not in weights:
            continue
        d = weights[l.name]
        if l.type == "Convolution":
            c = d["W"].shape[0]
            blobs = [d["W"], d["b"].reshape(c)]
        elif l.type == "BN":
            c = d["scale"].shape[0]
            blobs = [d[k].reshape(1, c, 1, 1) for k in ("scale", "shift", "mean", "var")]
        else:
            continue
        body = _enc_ld(1, l.name.encode()) + _enc_ld(2, l.type.encode()) + b"".join(_enc_ld(7, _enc_blob(b)) for b in blobs)
        out += _enc_ld(100, body)
    with open(path, "wb") as f:
        f.write(bytes(out))
