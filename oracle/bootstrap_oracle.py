"""CPU oracle for target bootstrapping: the reference's closed forms, restated.

TEST INFRASTRUCTURE ONLY: imported by tests/ (and nothing else).  The product computes the same targets on the
GPU from the Gram matrix of the selected rows (csrc/vq_boot.hip, Woodbury form); this file keeps the reference's own
explicit 1024x1024 inverses so that the two routes check each other.

Parity status: PINNED.  oracle/gen_golden_bootstrap.py runs the reference's ``TargetClip`` itself and records its
targets under tests/golden/bootstrap*.{json,npz}; tests/test_bootstrap_oracle.py checks every function here against
them.  Line numbers refer to src/models/target_clip.py of the reference.
"""
from __future__ import annotations

import random
from typing import Dict, List, Mapping, Sequence

import numpy as np


def random_fraction(n_items: int, fraction, replacement: bool) -> List[int]:
    """target_clip.py:297-309: positions to keep -- sample / choices, then ``list(set(...))`` (unique, set order)."""
    tmatches = round(n_items * fraction)
    tmatches = max(tmatches, 1)
    if replacement is False:
        tsamples = random.sample(range(n_items), tmatches)
    else:
        tsamples = random.choices(range(n_items), k=tmatches)
    return list(set(tsamples))


def bootstrap_valid(X_rows: np.ndarray) -> np.ndarray:
    """target_clip.py:192-197: rows = validated matches [m][D]; w = X (X^T X)^-1 1 with X = rows^T."""
    X = np.asarray(X_rows, dtype=np.float64).T
    M = np.matmul(X.T, X)
    M_inv = np.linalg.inv(M)
    mu = np.sum(M_inv, axis=1).reshape([-1, 1])
    return np.dot(X, mu).T[0]


def bootstrap_valid_invalid(X_rows: np.ndarray, Y_rows: np.ndarray, mu: float) -> np.ndarray:
    """target_clip.py:245-260 with X [m][D] validated matches, Y [n][D] validated non-matches."""
    X = np.asarray(X_rows, dtype=np.float64)
    Y = np.asarray(Y_rows, dtype=np.float64)
    tr_YYT = np.trace(np.matmul(Y, Y.T))
    scale = mu / tr_YYT
    M = np.eye(Y.shape[1]) + scale * np.matmul(Y.T, Y)
    M_inv = np.linalg.inv(M)
    B = np.matmul(X, np.matmul(M_inv, X.T))
    B_inv = np.linalg.inv(B)
    w_1 = np.matmul(np.matmul(M_inv, X.T), B_inv)
    w_2 = M_inv - np.matmul(np.matmul(w_1, X), M_inv)
    w_3 = np.sum(np.matmul(w_2, scale * Y.T), axis=1).reshape([-1, 1])
    w_final = w_3 + np.sum(w_1, axis=1).reshape([-1, 1])
    return w_final.T[0]


def _stack(dicts: Sequence[Mapping], streams, splits) -> Dict:
    """target_clip.py:186-190 / :233-242: per (stream, split) the list of feature vectors, in list order."""
    out = {st: {sp: [] for sp in splits} for st in streams}
    for fd in dicts:
        for st, split_features in fd.items():
            for sp, feature in split_features.items():
                out[st][sp].append(feature)
    return out


def dynamic_target_adjustment(valid: List[Mapping], invalid: List[Mapping], splits, streams, b_fraction, replacement: bool,
                              mu: float) -> Dict:
    """target_clip.py:84-104 (+ :161-198, :200-261): one new target {stream: {split: ndarray}}; draws from ``random``
    in the reference's order."""
    if invalid:
        valid = [valid[i] for i in random_fraction(len(valid), b_fraction, replacement)]
        invalid = [invalid[i] for i in random_fraction(len(invalid), b_fraction, replacement)]
        xf, yf = _stack(valid, streams, splits), _stack(invalid, streams, splits)
        return {st: {sp: bootstrap_valid_invalid(xf[st][sp], yf[st][sp], mu) for sp in splits} for st in streams}
    if b_fraction != 1 or replacement is True:
        valid = [valid[i] for i in random_fraction(len(valid), b_fraction, replacement)]
    xf = _stack(valid, streams, splits)
    return {st: {sp: bootstrap_valid(xf[st][sp]) for sp in splits} for st in streams}


def target_features(valid: List[Mapping], invalid: List[Mapping], splits, streams, bootstrap_type: str, mu: float,
                    f_bootstrap, f_memory: float, nbags: int, previous=None) -> Dict:
    """target_clip.py:41-73, cases 3-5 (the caller handles cases 1-2: no bootstrapping -> scaled reference clip)."""
    if bootstrap_type == "simple":
        t = dynamic_target_adjustment(valid, invalid, splits, streams, f_bootstrap, False, mu)
        return {st: {sp: t[st][sp] for sp in splits} for st in streams}
    if bootstrap_type == "partial_update":
        t = dynamic_target_adjustment(valid, invalid, splits, streams, f_bootstrap, False, mu)
        if previous:                                                                     # :75-82
            for st in streams:
                for sp in splits:
                    t[st][sp] = np.multiply(f_memory, t[st][sp]) + np.multiply((1 - f_memory), previous[st][sp])
        return t
    if bootstrap_type == "bagging":                                                      # :145-159
        bags = [dynamic_target_adjustment(valid, invalid, splits, streams, 1, True, mu) for _ in range(nbags)]
        return {st: {sp: np.average([bags[b][st][sp] for b in range(nbags)], axis=0) for sp in splits} for st in streams}
    raise Exception("Error: bootstrap_type should be one of 'simple', 'partial_update', or 'bagging'")


def woodbury_coefficients(G: np.ndarray, m: int, mu: float):
    """The product's route (csrc/vq_boot.hip), in numpy: from the Gram matrix G of the rows [X; Y] (m valid first)
    the coefficients (a, b) with  w = X^T a + Y^T b.  Algebraically equal to bootstrap_valid_invalid / bootstrap_valid:
        scale = mu / tr(Gyy);  K' = (I + scale Gyy)^-1;  C = scale K' Gyx;  gamma = scale K' 1
        B = Gxx - Gxy C;  beta = B^-1 1;  delta = B^-1 Gxy gamma;  a = beta - delta;  b = gamma - C a
    """
    G = np.asarray(G, dtype=np.float64)
    r = G.shape[0]
    n = r - m
    Gxx, Gxy, Gyy = G[:m, :m], G[:m, m:], G[m:, m:]
    scale = mu / np.trace(Gyy) if n > 0 and mu != 0 else 0.0
    if n == 0 or scale == 0.0:
        return np.linalg.solve(Gxx, np.ones(m)), np.zeros(n)
    Kc = np.linalg.solve(np.eye(n) + scale * Gyy, np.concatenate([Gxy.T, np.ones((n, 1))], axis=1))
    C, gamma = scale * Kc[:, :m], scale * Kc[:, m]
    B = Gxx - Gxy @ C
    sol = np.linalg.solve(B, np.stack([np.ones(m), Gxy @ gamma], axis=1))
    a = sol[:, 0] - sol[:, 1]
    return a, gamma - C @ a
