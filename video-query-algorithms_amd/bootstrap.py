"""Closed-form target bootstrapping on the GPU (the numeric core of the reference's "dynamic target adjustment").

``bootstrap_targets`` replaces the two matrix formulas of src/models/target_clip.py:
``_bootstrap_valid_matches`` (:192-197) and ``_bootstrap_valid_plus_invalid`` (:245-260).  The caller
(``target_clip.TargetClip``) keeps the reference's sampling, bagging and averaging logic on the host; every
(stream, split[, bag]) problem of one round goes to the device in a single launch (csrc/vq_boot.hip).
"""
from __future__ import annotations

import ctypes as C
from typing import Sequence, Tuple

import numpy as np

from . import _lib
from ._lib import call


def bootstrap_targets(problems: Sequence[Tuple[np.ndarray, np.ndarray | None]], mu: float, device: int = 0) -> np.ndarray:
    """problems: (X [m][D] validated matches, Y [n][D] validated non-matches or None) per (stream, split[, bag]).
    Returns the new targets [len(problems)][D] fp64.  Raises ``VqError`` if a system is singular (the reference's
    ``np.linalg.inv`` raises ``LinAlgError`` there)."""
    if not problems:
        raise ValueError("no problems")
    dim = np.asarray(problems[0][0]).shape[-1]
    all32 = all(np.asarray(x).dtype == np.float32 and (y is None or np.asarray(y).dtype == np.float32) for x, y in problems)
    dt = np.float32 if all32 else np.float64
    blocks, nv, ni = [], [], []
    for x, y in problems:
        x = np.asarray(x, dtype=dt).reshape(-1, dim)
        y = np.zeros((0, dim), dtype=dt) if y is None else np.asarray(y, dtype=dt).reshape(-1, dim)
        blocks += [x, y]
        nv.append(x.shape[0])
        ni.append(y.shape[0])
    rows = np.ascontiguousarray(np.concatenate(blocks, axis=0))
    nv = np.ascontiguousarray(nv, dtype=np.int32)
    ni = np.ascontiguousarray(ni, dtype=np.int32)
    out = np.empty((len(problems), dim), dtype=np.float64)
    call("vq_bootstrap_targets", rows.ctypes.data_as(C.c_void_p), _lib.VQ_F32 if all32 else _lib.VQ_F64, len(problems),
         nv.ctypes.data_as(C.c_void_p), ni.ctypes.data_as(C.c_void_p), dim, float(mu), device, out.ctypes.data_as(C.c_void_p))
    return out
