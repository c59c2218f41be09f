"""CPU oracle for hot path B (similarity scoring / ranking / weight update).

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is imported by the
product package; only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may use it, and only as the checker.

Parity status: PINNED.  ``oracle/gen_golden.py`` imports the reference's own
``models`` package in the build container and records its outputs under
``tests/golden/``; ``tests/test_oracle_sim.py`` checks every function here
against those vectors.

Two restatements of the same arithmetic live here:

* ``faithful_*``  -- pure-Python loops over the nested dict / list inputs the
  REST API hands the reference (small cases; also the ``"port"`` CPU baseline).
* ``dense_*``     -- vectorised numpy fp64 over dense ``[N,S,E,D]`` arrays
  (the checker used at sizes where Python loops take too long).

Each function cites the reference lines it follows (paths relative to the
reference checkout, ``src/models/...``).
"""
from __future__ import annotations

import random
from typing import Dict, Iterable, List, Mapping, Sequence, Tuple

import numpy as np


# ---------------------------------------------------------------------------
# B1  target scaling -- target_clip.py:311-313, :137-143, :263-286
# ---------------------------------------------------------------------------
def scale_feature(f) -> np.ndarray:
    """``t = f / (f . f)`` (target_clip.py:311-313)."""
    f = np.asarray(f, dtype=np.float64)
    return f / np.dot(f, f)


def faithful_clip_features(records: Iterable[Mapping], streams: Sequence[str],
                           feature_name: str):
    """target_clip.py:263-286: API records of ONE clip -> ({stream:{split:vec}}, splits)."""
    results = {s: {} for s in streams}
    splits = set()
    for rec in records:
        stream_type = rec["dnn_stream_id"]
        if stream_type in streams and rec["name"] == feature_name:
            fsplit = rec["dnn_stream_split"]
            splits.add(fsplit)
            results[stream_type][fsplit] = rec["feature_vector"]
    return results, splits


def faithful_scaled_ref_clip_features(ref_clip_features):
    """target_clip.py:137-143."""
    out = {}
    for stream, split_features in ref_clip_features.items():
        out[stream] = {}
        for split, feature in split_features.items():
            out[stream][split] = scale_feature(feature).tolist()
    return out


# ---------------------------------------------------------------------------
# B2  candidate regrouping -- ticket.py:358-382
# ---------------------------------------------------------------------------
def faithful_candidate_features(records: Iterable[Mapping], splits, streams,
                                feature_name: str):
    """ticket.py:358-382: search-set records -> {stream:{split:{clip:vec}}}."""
    cand = {}
    for stream in streams:
        cand[stream] = {}
        for split in splits:
            cand[stream][split] = {}
    for tf in records:
        if (tf["dnn_stream_id"] in streams and tf["name"] == feature_name
                and tf["dnn_stream_split"] in splits):
            cand[tf["dnn_stream_id"]][tf["dnn_stream_split"]][tf["video_clip_id"]] = tf["feature_vector"]
    return cand


# ---------------------------------------------------------------------------
# B3  similarities -- ticket.py:120-163
# ---------------------------------------------------------------------------
def faithful_similarities(target_features, candidates):
    """ticket.py:144-160.  Returns {clip:{stream:[avg, n_e]}} in first-seen order."""
    avgd = {}
    for stream_type, all_splits in target_features.items():
        sims = {}
        for split, target_feature in all_splits.items():
            for clip, cand in candidates[stream_type][split].items():
                s = np.dot(target_feature, cand)
                sims[clip] = sims.get(clip, []) + [s]
        for clip_id, arr in sims.items():
            n = len(arr)
            avgd[clip_id] = avgd.get(clip_id, {})
            avgd[clip_id].update({stream_type: [sum(arr) / n, n]})
    return avgd


def dense_similarities(feats: np.ndarray, target: np.ndarray,
                       present: np.ndarray | None = None):
    """Vectorised ticket.py:151,155-160 on dense arrays.

    feats   [N,S,E,D] (any float dtype; promoted to fp64 like ``np.dot`` on lists)
    target  [S,E,D] fp64
    present [N,S,E] bool/u8 or None (dense)
    Returns (sims [N,S,E] fp64, avg [N,S] fp64, n_e [N,S] int32).  Where
    ``n_e == 0`` the average is NaN (the reference would have no entry).
    The ensemble sum runs sequentially over e exactly like Python's ``sum``.
    """
    x = np.asarray(feats)
    t = np.asarray(target, dtype=np.float64)
    n, s, e, d = x.shape
    sims = np.empty((n, s, e), dtype=np.float64)
    # chunk to bound the fp64 temporary
    step = max(1, (1 << 24) // (s * e * d))
    for i in range(0, n, step):
        sims[i:i + step] = np.einsum("nsed,sed->nse", x[i:i + step].astype(np.float64), t)
    if present is None:
        pres = np.ones((n, s, e), dtype=bool)
    else:
        pres = np.asarray(present).astype(bool)
    acc = np.zeros((n, s), dtype=np.float64)
    for k in range(e):                       # sum([...]) is sequential, starts at int 0
        acc = np.where(pres[:, :, k], acc + sims[:, :, k], acc)
    n_e = pres.sum(axis=2).astype(np.int32)
    with np.errstate(invalid="ignore", divide="ignore"):
        avg = acc / n_e
    return sims, avg, n_e


# ---------------------------------------------------------------------------
# B4  scores -- ticket.py:165-180
# ---------------------------------------------------------------------------
def faithful_scores(similarities, weights):
    """ticket.py:172-180."""
    scores = {}
    for clip, vsim in similarities.items():
        ssum = 0
        denom = 0
        for stream_type, w in weights.items():
            ssum += (w * (1 - vsim[stream_type][0])) ** 2
            denom += w ** 2
        scores[clip] = 1 - np.sqrt(ssum / denom)
    return scores


def dense_scores(avg: np.ndarray, w: Sequence[float]) -> np.ndarray:
    """ticket.py:172-180 on avg[N,S] with correctly rounded squares (``x*x``).

    The reference squares with ``**2`` on numpy float64 SCALARS, which calls libm ``pow(x, 2.0)``;
    glibc's pow is not always correctly rounded, so about 1 score in 5 000 differs from this
    function by one ulp of the sum (<= 2.3e-16 absolute).  ``dense_scores_libm`` reproduces the
    reference bit for bit (slow scalar loop); the HIP kernel is bit-identical to THIS function.
    """
    avg = np.asarray(avg, dtype=np.float64)
    ssum = np.zeros(avg.shape[0], dtype=np.float64)
    denom = 0.0
    for s, ws in enumerate(w):
        ws = float(ws)
        term = ws * (1.0 - avg[:, s])
        ssum = ssum + term * term
        denom = denom + ws * ws
    return 1.0 - np.sqrt(ssum / denom)


def dense_scores_libm(avg: np.ndarray, w: Sequence[float]) -> np.ndarray:
    """ticket.py:172-180 with the reference's own scalar operations (numpy-scalar ``**``)."""
    avg = np.asarray(avg, dtype=np.float64)
    out = np.empty(avg.shape[0], dtype=np.float64)
    ws = [np.float64(x) for x in w]
    for i in range(avg.shape[0]):
        ssum = 0
        denom = 0
        for s, wv in enumerate(ws):
            ssum += (wv * (1 - avg[i, s])) ** 2
            denom += wv ** 2
        out[i] = 1 - np.sqrt(ssum / denom)
    return out


# ---------------------------------------------------------------------------
# B5  selection -- ticket.py:311-356
# ---------------------------------------------------------------------------
def faithful_select(scores: Mapping, ref_clip_id, user_matches: Mapping,
                    threshold=0.8, max_number_matches=20, near_miss=0.5, rng=random):
    """ticket.py:325-355.  ``rng`` is the ``random`` module (global state) by default."""
    lower_limit = threshold - near_miss * (1 - threshold)
    match_c = {k: v for k, v in scores.items() if v >= threshold}
    near_c = {k: v for k, v in scores.items() if lower_limit <= v < threshold}
    mscores = int(min(max_number_matches / 2, len(match_c)))
    m_near = int(min(max_number_matches - mscores, len(near_c)))
    # CPython <= 3.10 turns a dict view into tuple(view) inside random.sample
    match_scores = rng.sample(list(match_c.items()), mscores)
    near_max = {}
    if m_near > 0:
        m_near -= 1
        kmax = max(near_c, key=lambda key: near_c[key])
        near_max = {kmax: scores[kmax]}
        near_c.pop(kmax)
    near_scores = rng.sample(list(near_c.items()), m_near)
    matches = dict(match_scores + near_scores)
    matches.update(near_max)
    prev = {ref_clip_id: scores[ref_clip_id]} if ref_clip_id in scores else {}
    if user_matches:
        for clip, value in user_matches.items():
            if value is True:
                prev.update({int(clip): scores[int(clip)]})
    matches.update(prev)
    return matches


def dense_select_partition(scores: np.ndarray, threshold: float, near_miss: float):
    """The deterministic half of ticket.py:325-340 on an array.

    Returns (match_rows, near_rows, near_argmax_row) with rows in original order;
    near_argmax_row = first row attaining the max of the near band, or -1.
    """
    v = np.asarray(scores, dtype=np.float64)
    lower = threshold - near_miss * (1 - threshold)
    match_rows = np.flatnonzero(v >= threshold)
    near_rows = np.flatnonzero((lower <= v) & (v < threshold))
    amax = int(near_rows[np.argmax(v[near_rows])]) if near_rows.size else -1
    return match_rows, near_rows, amax


# ---------------------------------------------------------------------------
# B6  lowest scoring user match -- ticket.py:301-309
# ---------------------------------------------------------------------------
def faithful_lowest_scoring_user_match(scores: Mapping, user_matches: Mapping):
    """ticket.py:301-309 (returns the LAST true-match clip id, not the argmin)."""
    min_score = 1
    min_clip = None
    for clip, score in scores.items():
        if str(clip) in user_matches:
            if user_matches[str(clip)] is True:
                min_score = min(min_score, score)
                min_clip = clip
    return min_score, min_clip


# ---------------------------------------------------------------------------
# B7  weight update -- hyperparameter.py:29-114
# ---------------------------------------------------------------------------
WEIGHT_GRID = np.arange(0.5, 2.5, 0.05)      # hyperparameter.py:20
THRESHOLD_GRID = np.arange(0.5, 1.1, 0.02)   # hyperparameter.py:21


def quad_fit(x, y):
    """hyperparameter.py:85-114 (5-point separable parabola, clamp, verify)."""
    w0 = (y[4] - y[0]) * x[0][1] ** 2 + (y[2] - y[4]) * x[0][0] ** 2 - (y[2] - y[0]) * x[0][2] ** 2
    w0 = 0.5 * w0 / ((y[4] - y[0]) * x[0][1] + (y[2] - y[4]) * x[0][0] - (y[2] - y[0]) * x[0][2])
    a0 = (y[2] - y[0]) / ((x[0][1] - w0) ** 2 - (x[0][0] - w0) ** 2)
    th0 = (y[3] - y[1]) * x[1][1] ** 2 + (y[2] - y[3]) * x[1][0] ** 2 - (y[2] - y[1]) * x[1][2] ** 2
    th0 = 0.5 * th0 / ((y[3] - y[1]) * x[1][1] + (y[2] - y[3]) * x[1][0] - (y[2] - y[1]) * x[1][2])
    b0 = (y[2] - y[1]) / ((x[1][1] - th0) ** 2 - (x[1][0] - th0) ** 2)
    c0 = y[2] - a0 * (x[0][1] - w0) ** 2 - b0 * (x[1][1] - th0) ** 2
    w0 = min(w0, x[0][2]); w0 = max(w0, x[0][0])
    th0 = min(th0, x[1][2]); th0 = max(th0, x[1][0])
    eps = 10 ** -6
    y0 = a0 * (x[0][0] - w0) ** 2 + b0 * (x[1][1] - th0) ** 2 + c0
    y1 = a0 * (x[0][1] - w0) ** 2 + b0 * (x[1][0] - th0) ** 2 + c0
    y2 = a0 * (x[0][1] - w0) ** 2 + b0 * (x[1][1] - th0) ** 2 + c0
    y3 = a0 * (x[0][1] - w0) ** 2 + b0 * (x[1][2] - th0) ** 2 + c0
    y4 = a0 * (x[0][2] - w0) ** 2 + b0 * (x[1][1] - th0) ** 2 + c0
    if (abs(y[0] - y0) + abs(y[1] - y1) + abs(y[2] - y2) + abs(y[3] - y3) + abs(y[4] - y4)) > eps:
        w0 = x[0][1]
        th0 = x[1][1]
    return w0, th0


def faithful_optimize_weights(similarities, matches: List[Mapping], streams, ballast, eps_threshold,
                              weight_grid=WEIGHT_GRID, threshold_grid=THRESHOLD_GRID):
    """hyperparameter.py:45-76.  Returns (weights dict, threshold, losses[40,31], last_scores)."""
    match_status = {}
    for m in matches:
        if m["user_match"] is not None:
            match_status[m["video_clip"]] = m["user_match"]
        else:
            match_status[m["video_clip"]] = m["is_match"]
    losses = 100 * np.ones([weight_grid.shape[0], threshold_grid.shape[0]])
    scores = {}
    for iw, w in enumerate(weight_grid):
        scores = faithful_scores(similarities, {streams[0]: 1.0, streams[1]: w})
        for ith, th in enumerate(threshold_grid):
            loss = 0.5 * th
            for clip in match_status:
                sc = scores[clip]
                loss += (np.heaviside(sc - th, 1) - match_status[clip]) * (sc - th) \
                    * (1 + match_status[clip] * ballast)
            losses[iw, ith] = loss / len(match_status)
    iw0, ith0 = np.unravel_index(np.argmin(losses, axis=None), losses.shape)
    if iw0 == 0 or ith0 == 0 or iw0 == len(weight_grid) - 1 or ith0 == len(threshold_grid) - 1:
        w_opt = weight_grid[iw0]
        th_opt = threshold_grid[ith0]
    else:
        xr = [(weight_grid[iw0 - 1], weight_grid[iw0], weight_grid[iw0 + 1]),
              (threshold_grid[ith0 - 1], threshold_grid[ith0], threshold_grid[ith0 + 1])]
        yd = [losses[iw0 - 1, ith0], losses[iw0, ith0 - 1], losses[iw0, ith0], losses[iw0, ith0 + 1],
              losses[iw0 + 1, ith0]]
        w_opt, th_opt = quad_fit(xr, yd)
    return {streams[0]: 1.0, streams[1]: w_opt}, th_opt - eps_threshold, losses, scores


def dense_loss_grid(grid_scores: np.ndarray, labels: np.ndarray, ballast: float,
                    threshold_grid=THRESHOLD_GRID) -> np.ndarray:
    """hyperparameter.py:56-65 given grid_scores[G,L] and labels[L] (bool/0-1).

    Sequential accumulation over the L labelled clips (same order as the dict walk).
    """
    g, l = grid_scores.shape
    th = threshold_grid[None, :]
    loss = np.broadcast_to(0.5 * th, (g, threshold_grid.shape[0])).copy()
    for j in range(l):
        y = float(labels[j])
        d = grid_scores[:, j][:, None] - th
        loss = loss + (np.heaviside(d, 1) - y) * d * (1 + y * ballast)
    return loss / l


# ---------------------------------------------------------------------------
# B8  finalize near-miss formula -- compute_matches.py:82-84
# ---------------------------------------------------------------------------
def finalize_near_miss(threshold: float, low_score: float, compute_eps: float) -> float:
    return max(threshold - low_score, 0) / max(1 - threshold, compute_eps)


# ---------------------------------------------------------------------------
# ranking helpers (final report order, ticket.py:266: stable sort, descending)
# ---------------------------------------------------------------------------
def dense_topk(scores: np.ndarray, k: int) -> Tuple[np.ndarray, np.ndarray]:
    """Rows of the k largest scores, descending, ties by original order (stable)."""
    v = np.asarray(scores, dtype=np.float64)
    order = np.argsort(-v, kind="stable")[:k]
    return order.astype(np.int64), v[order]


# ---------------------------------------------------------------------------
# synthetic workloads; shared by tests, smoke and bench
# ---------------------------------------------------------------------------
def cfg1_features(n=10000, e=3, d=1024, seed=0) -> np.ndarray:
    """SURVEY 8(d) cfg 1: |N(0,1)| fp32, x3.8 (rgb) / x1.2 (flow); returns [N,S=2,E,D] fp32."""
    rng = np.random.default_rng(seed)
    x = np.abs(rng.standard_normal((e, 2, n, d), dtype=np.float32))
    x[:, 0] *= np.float32(3.8)
    x[:, 1] *= np.float32(1.2)
    return np.ascontiguousarray(x.transpose(2, 1, 0, 3))


_M64 = (1 << 64) - 1


def synth_hash_u24(seed: int, idx: np.ndarray) -> np.ndarray:
    """Counter-based generator shared with the device kernel ``vq_db_generate``.

    splitmix64 finaliser of (idx + seed*0x9E3779B97F4A7C15); returns the top 24 bits.
    """
    z = (idx.astype(np.uint64) + np.uint64((seed * 0x9E3779B97F4A7C15) & _M64))
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(40)).astype(np.uint32)


def synth_features(seed: int, row0: int, nrows: int, s: int, e: int, d: int,
                   scales: Sequence[float]) -> np.ndarray:
    """Rows [row0,row0+nrows) of the on-device synthetic DB: value = u24 * 2^-24 * scale[s]."""
    per = s * e * d
    idx = (np.arange(row0 * per, (row0 + nrows) * per, dtype=np.uint64))
    with np.errstate(over="ignore"):
        u = synth_hash_u24(seed, idx)
    x = (u.astype(np.float32) * np.float32(2.0 ** -24)).reshape(nrows, s, e, d)
    sc = np.asarray(scales, dtype=np.float32).reshape(1, s, 1, 1)
    return x * sc
