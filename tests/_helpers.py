"""Shared test helpers (no product code, no oracle code)."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
STREAMS = ("rgb", "warped_optical_flow")
DEFAULT_WEIGHTS = {"rgb": 1.0, "warped_optical_flow": 1.5}
SEED = "73459912436"


def golden_json(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def golden_npy(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def records_from_dense(x, clip_ids, splits, present=None, order="split_major", extra=()):
    """[N,S,E,D] -> API feature records (the wire format of ticket.py:374-381)."""
    n, s, e, d = x.shape
    recs = []
    if order == "split_major":
        it = ((si, ei, ci) for ei in range(e) for si in range(s) for ci in range(n))
    else:
        it = ((si, ei, ci) for ci in range(n) for si in range(s) for ei in range(e))
    for si, ei, ci in it:
        if present is not None and not present[ci, si, ei]:
            continue
        recs.append({"dnn_stream_id": STREAMS[si], "dnn_stream_split": splits[ei], "name": "global_pool",
                     "video_clip_id": int(clip_ids[ci]), "feature_vector": x[ci, si, ei].astype(np.float64).tolist()})
    return recs + list(extra)


RAGGED_EXTRA = [
    {"dnn_stream_id": "rgb", "dnn_stream_split": 1, "name": "fc-action", "video_clip_id": 999,
     "feature_vector": [1.0] * 1024},
    {"dnn_stream_id": "audio", "dnn_stream_split": 1, "name": "global_pool", "video_clip_id": 998,
     "feature_vector": [1.0] * 1024},
    {"dnn_stream_id": "rgb", "dnn_stream_split": 9, "name": "global_pool", "video_clip_id": 997,
     "feature_vector": [1.0] * 1024},
]


def golden_target_array(g, splits=(1, 2, 3)):
    return np.array([[g["target"][st][str(sp)] for sp in splits] for st in STREAMS], dtype=np.float64)


def mean_sum_bits(values, T):
    """For positive float64 values a: the fewest significant bits B of a number s with fl(s / T) == a, s taken as T * a rounded to B
    bits (53 when nothing shorter reproduces a).  When a = (fp64 sum of T fp32 numbers) / T -- calcSig_wOF.py:82,
    ``np.array(frame_features).mean(axis=0)`` over T float32 blobs -- that sum is exact in fp64 and short: 24 bits + log2(T) +
    the exponent spread of the T addends (about 31 bits in the median); for any other double, or another T, B is 53."""
    a = np.asarray(values, dtype=np.float64).reshape(-1)
    s = a * float(T)
    m, e = np.frexp(s)
    need = np.full(a.shape, 53, dtype=np.int64)
    for bits in range(52, 9, -1):
        r = np.ldexp(np.round(m * 2.0 ** bits), e - bits)
        need[r / float(T) == a] = bits
    return need
