"""Binary feature store: the resident-database layout on disk.

The reference moves a search set's features as one JSON GET of API records (``Ticket._get_candidate_features``,
src/models/ticket.py:358-382: ``{dnn_stream_id, dnn_stream_split, name, video_clip_id, feature_vector}`` per
(clip, stream, split)) after ``load_db`` parsed the CSV tree (src/api/api_load_records.py:41-61).  At the 1M-clip
scale of BASELINE configs[3] that is 10 G floats of JSON.  The store keeps the same information as

    <dir>/features.npy   [N][S][E][D] float32 (or float64), C order -- the FeatureDB block, memory-mappable
    <dir>/clip_ids.npy   [N] int64, row order = the order the reference would first meet the clips
    <dir>/present.npy    [N][S][E] uint8 (only when some (clip, stream, split) is missing)
    <dir>/meta.json      {"streams": [...], "splits": [...], "feature_name": "global_pool", "dim": D}

and ``FeatureDB.from_store`` streams it to the GPU in bounded chunks (any row range: one shard per rank).
``store_from_csv_tree`` builds a store from the reference's ``data/features`` layout.
"""
from __future__ import annotations

import json
import os
from typing import Dict, Sequence

import numpy as np

from .tsn import feature_csv


def save_store(path: str, feats: np.ndarray, clip_ids: Sequence[int], streams: Sequence[str], splits: Sequence[int],
               feature_name: str = "global_pool", present: np.ndarray | None = None) -> str:
    feats = np.asarray(feats)
    if feats.ndim != 4 or feats.dtype not in (np.float32, np.float64):
        raise ValueError("feats must be [N,S,E,D] float32/float64")
    n, s, e, d = feats.shape
    clip_ids = np.asarray(clip_ids, dtype=np.int64)
    if clip_ids.shape != (n,) or len(streams) != s or len(splits) != e:
        raise ValueError("clip_ids / streams / splits do not match the feature block")
    if len(set(clip_ids.tolist())) != n:
        raise ValueError("clip ids must be unique")
    os.makedirs(path, exist_ok=True)
    np.save(os.path.join(path, "features.npy"), np.ascontiguousarray(feats))
    np.save(os.path.join(path, "clip_ids.npy"), clip_ids)
    ppath = os.path.join(path, "present.npy")
    if present is not None and not np.asarray(present).all():
        p = np.asarray(present).astype(np.uint8)
        if p.shape != (n, s, e):
            raise ValueError("present must be [N,S,E]")
        np.save(ppath, p)
    elif os.path.exists(ppath):
        os.remove(ppath)
    with open(os.path.join(path, "meta.json"), "w") as f:
        json.dump({"streams": list(streams), "splits": [int(x) for x in splits], "feature_name": feature_name, "dim": int(d),
                   "n": int(n), "dtype": str(feats.dtype)}, f)
    return path


def open_store(path: str):
    """(meta, features memmap [N,S,E,D], clip_ids [N], present or None) -- nothing is read until it is sliced."""
    with open(os.path.join(path, "meta.json")) as f:
        meta = json.load(f)
    feats = np.load(os.path.join(path, "features.npy"), mmap_mode="r", allow_pickle=False)
    ids = np.load(os.path.join(path, "clip_ids.npy"), allow_pickle=False)
    ppath = os.path.join(path, "present.npy")
    present = np.load(ppath, mmap_mode="r", allow_pickle=False) if os.path.exists(ppath) else None
    if feats.ndim != 4 or feats.shape[0] != ids.shape[0] or feats.shape[1] != len(meta["streams"]) or feats.shape[2] != len(meta["splits"]):
        raise ValueError("feature store %s is inconsistent" % path)
    return meta, feats, ids, present


def store_from_csv_tree(features_dir: str, out: str, streams: Sequence[str] = feature_csv.STREAM_MODES, clip_id_base: int = 0,
                        dtype=np.float32) -> str:
    """``<features_dir>/<video>/<name ending in the split digit>/<stream>_<blob>_features.csv`` (the layout
    ``load_db`` walks, src/api/load_db.py:10-28 + api_load_records.py:41-61) -> a store.  Clips of successive videos
    (sorted by name) get ids ``clip_id_base + running index + 1`` in (video, clip number) order; a (clip, stream, split)
    without a row is marked absent."""
    per_video = []
    all_splits = set()
    for video in sorted(d for d in os.listdir(features_dir) if os.path.isdir(os.path.join(features_dir, d))):
        vdir = os.path.join(features_dir, video)
        by_split: Dict[int, dict] = {}
        for sd in sorted(d for d in os.listdir(vdir) if os.path.isdir(os.path.join(vdir, d))):
            nsplit, per_stream = feature_csv.read_split_dir(os.path.join(vdir, sd))
            by_split[nsplit] = per_stream
            all_splits.add(nsplit)
        clips = sorted({int(c) for ps in by_split.values() for (cl, _f, _m) in ps.values() for c in cl})
        per_video.append((video, clips, by_split))
    splits = sorted(all_splits)
    n = sum(len(c) for _v, c, _b in per_video)
    if n == 0:
        raise ValueError("no feature files under %s" % features_dir)
    dim = None
    for _v, _c, by_split in per_video:
        for ps in by_split.values():
            for (_cl, f, _m) in ps.values():
                dim = f.shape[1]
    feats = np.zeros((n, len(streams), len(splits), dim), dtype=dtype)
    present = np.zeros((n, len(streams), len(splits)), dtype=np.uint8)
    ids = np.zeros(n, dtype=np.int64)
    names = []
    row0 = 0
    for video, clips, by_split in per_video:
        local = {c: row0 + i for i, c in enumerate(clips)}
        for i, c in enumerate(clips):
            ids[row0 + i] = clip_id_base + row0 + i + 1
            names.append((video, c))
        for ei, sp in enumerate(splits):
            for si, st in enumerate(streams):
                if sp in by_split and st in by_split[sp]:
                    cl, f, _m = by_split[sp][st]
                    rows = np.array([local[int(c)] for c in cl], dtype=np.int64)
                    feats[rows, si, ei] = f.astype(dtype)
                    present[rows, si, ei] = 1
        row0 += len(clips)
    save_store(out, feats, ids, streams, splits, present=present)
    with open(os.path.join(out, "clips.json"), "w") as f:
        json.dump([{"id": int(i), "video": v, "clip": int(c)} for i, (v, c) in zip(ids, names)], f)
    return out
