"""Drop-in for the scoring methods of the reference's ``Ticket`` (src/models/ticket.py).

``TicketScoring`` carries the four hot-path methods with the reference's names, arguments and
result attributes -- ``compute_similarities`` (ticket.py:120-163), ``compute_scores``
(ticket.py:165-180), ``select_clips_to_review`` (ticket.py:311-356) and
``lowest_scoring_user_match`` (ticket.py:301-309) -- running on a resident :class:`FeatureDB`
through the HIP kernels.  The REST / reporting methods of the reference class are untouched: mix
this class in front of it (``class Ticket(TicketScoring, ReferenceTicket)``) or call
:func:`install` on the reference's classes.  ``Ticket`` below is a self-contained
variant for offline use whose ``_request`` answers from in-memory records.
"""
from __future__ import annotations

import random
from collections.abc import Mapping

import numpy as np

from .feature_db import FeatureDB


class ScoreMap(Mapping):
    """``{video_clip_id: score}`` backed by the arrays read back from the device; iteration order is
    the order in which the reference would have inserted the clips (ticket.py:146-160)."""

    def __init__(self, clip_ids: np.ndarray, values: np.ndarray, row_of):
        self._ids, self._vals, self._row_of = clip_ids, values, row_of

    def __getitem__(self, clip):
        return self._vals[self._row_of(clip)]

    def __iter__(self):
        return iter(self._ids.tolist())

    def __len__(self):
        return self._ids.shape[0]

    def __contains__(self, clip):
        try:
            self._row_of(clip)
            return True
        except (KeyError, TypeError, ValueError):
            return False

    def items(self):
        return _Pairs(self._ids, self._vals)

    def values(self):
        return list(self._vals)


class _Pairs:
    def __init__(self, ids, vals):
        self._ids, self._vals = ids, vals

    def __iter__(self):
        return zip(self._ids.tolist(), self._vals)

    def __len__(self):
        return self._ids.shape[0]


class SimilarityMap(Mapping):
    """``{video_clip_id: {stream: [avg similarity, ensemble size]}}`` (ticket.py:123-124)."""

    def __init__(self, clip_ids, streams, avg, n_e, row_of):
        self._ids, self._streams, self._avg, self._ne, self._row_of = clip_ids, streams, avg, n_e, row_of

    def _entry(self, row):
        return {st: [self._avg[row, s], int(self._ne[row, s])]
                for s, st in enumerate(self._streams) if self._ne[row, s] > 0}

    def __getitem__(self, clip):
        return self._entry(self._row_of(clip))

    def __iter__(self):
        return iter(self._ids.tolist())

    def __len__(self):
        return self._ids.shape[0]

    def items(self):
        return ((int(c), self._entry(r)) for r, c in enumerate(self._ids.tolist()))


class TicketScoring:
    """Hot-path methods of the reference's Ticket, on the GPU.  Attributes read: ``target``
    (``target_features``, ``splits``), ``search_set``, ``ref_clip_id``, ``user_matches``; attributes
    written: ``similarities``, ``scores``, ``matches`` -- as in the reference."""

    feature_db: FeatureDB | None = None      # a resident DB may be attached up front
    feature_db_dtype = np.float64            # dtype used when the DB is built from API records
    device = 0

    # -- ticket.py:120-163 ------------------------------------------------------------------
    def compute_similarities(self, hyperparameters):
        target_features = self.target.target_features
        db = self.feature_db
        if db is None:
            records = self._request(["search-sets", "features"], {"id": self.search_set})   # ticket.py:363-365
            db = FeatureDB.from_records(records, target_features, hyperparameters.streams,
                                        hyperparameters.feature_name, dtype=self.feature_db_dtype,
                                        device=self.device)
            self.feature_db = db
        stream_names = getattr(db, "stream_names", None) or list(target_features.keys())
        slot_splits = getattr(db, "slot_splits", None) or [list(target_features[st].keys()) for st in stream_names]
        t = np.zeros((db.S, db.E, db.D), dtype=np.float64)
        slot_used = np.zeros((db.S, db.E), dtype=bool)
        for s, st in enumerate(stream_names):
            for e, sp in enumerate(slot_splits[s]):
                if st in target_features and sp in target_features[st]:
                    t[s, e] = np.asarray(target_features[st][sp], dtype=np.float64)
                    slot_used[s, e] = True
        if not slot_used.all():
            # the target lacks this (stream, split): the reference never visits it (ticket.py:146-148)
            base = db.present if db.present is not None else np.ones((db.n, db.S, db.E), dtype=np.uint8)
            db.set_present(base * slot_used[None].astype(np.uint8))
        db.set_query(t)
        db.scan(weights=None)
        avg, n_e = db.similarities()
        self._stream_names = stream_names
        self._avg, self._n_e = avg, n_e
        self.similarities = SimilarityMap(db.clip_ids, stream_names, avg, n_e, db.row_of)

    # -- ticket.py:165-180 ------------------------------------------------------------------
    def compute_scores(self, weights):
        db = self.feature_db
        w = np.zeros(db.S, dtype=np.float64)
        for stream_type, ws in weights.items():
            s = self._stream_names.index(stream_type) if stream_type in self._stream_names else -1
            if s < 0 or (self._n_e[:, s] == 0).any():
                raise KeyError(stream_type)          # vsim[stream_type] at ticket.py:177
            w[s] = ws
        db.rescore(w)
        self._score_values = db.scores()
        self.scores = ScoreMap(db.clip_ids, self._score_values, db.row_of)

    # -- ticket.py:301-309 ------------------------------------------------------------------
    def lowest_scoring_user_match(self):
        db = self.feature_db
        rows = sorted(db.row_of(int(c)) for c, v in self.user_matches.items()
                      if v is True and db.has_clip(c) and str(int(c)) == c)
        min_score, min_clip = 1, None
        for r in rows:                               # same order as the walk over self.scores
            min_score = min(min_score, self._score_values[r])
            min_clip = int(db.clip_ids[r])
        return min_score, min_clip

    # -- ticket.py:311-356 ------------------------------------------------------------------
    def select_clips_to_review(self, threshold=0.8, max_number_matches=20, near_miss=0.5):
        db = self.feature_db
        vals, ids = self._score_values, db.clip_ids
        lower_limit = threshold - near_miss * (1 - threshold)
        match_rows, near_rows, near_argmax = db.select(threshold, lower_limit)     # stable partition on the GPU
        mscores = int(min(max_number_matches / 2, len(match_rows)))
        m_near_scores = int(min(max_number_matches - mscores, len(near_rows)))
        # random.sample draws positions from (len(population), k) only, so sampling positions
        # consumes the generator exactly like sampling the (clip, score) pairs (ticket.py:333,341)
        picked = [match_rows[j] for j in random.sample(range(len(match_rows)), mscores)]
        near_max_row = None
        if m_near_scores > 0:
            m_near_scores -= 1
            near_max_row = near_argmax                                                 # first max in order
            near_rows = near_rows[near_rows != near_argmax]
        picked += [near_rows[j] for j in random.sample(range(len(near_rows)), m_near_scores)]
        matches = {int(ids[r]): vals[r] for r in picked}
        if near_max_row is not None:
            matches[int(ids[near_max_row])] = vals[near_max_row]
        # forced inclusions (ticket.py:346-356)
        if self.ref_clip_id in self.scores:
            previous_user_evals = {self.ref_clip_id: self.scores[self.ref_clip_id]}
        else:
            previous_user_evals = {}
        if self.user_matches:
            for clip, value in self.user_matches.items():
                if value is True:
                    previous_user_evals.update({int(clip): self.scores[int(clip)]})
        matches.update(previous_user_evals)
        self.matches = matches


class Ticket(TicketScoring):
    """Self-contained ticket for offline use: the attributes of ticket.py:38-57 plus an in-memory
    ``_request`` for the two feature endpoints the hot path calls."""

    def __init__(self, update_object, records=None, feature_db: FeatureDB | None = None, device: int = 0):
        self.query_id = update_object.get("query_id")
        self.video_id = update_object.get("video_id")
        self.ref_clip = update_object.get("ref_clip")
        self.ref_clip_id = update_object.get("ref_clip_id")
        self.search_set = update_object.get("search_set")
        self.number_of_matches_to_review = update_object.get("number_of_matches_to_review", 20)
        self.dynamic_target_adjustment = update_object.get("dynamic_target_adjustment", False)
        self.latest_query_result = update_object.get("latest_query_result")
        self.matches = update_object.get("matches", [])
        self.user_matches = update_object.get("user_matches", {})
        self.target = None
        self.similarities = {}
        self.scores = {}
        self.client = None
        self.schema = None
        self.feature_db = feature_db
        self.device = device
        self._records = records if records is not None else []
        # matches of the latest query result ({"video_clip": id, "user_match": True/False/None}), served in pages like
        # the API endpoint ["matches", "list"] that target bootstrapping reads (target_clip.py:114-121)
        self._match_list = update_object.get("match_list", [])
        self._match_page = int(update_object.get("match_page_size", 100))

    def _request(self, action, params):
        if action == ["search-sets", "features"]:
            return self._records
        if action == ["video-clips", "features"]:
            return [r for r in self._records if r["video_clip_id"] == params["id"]]
        if action == ["matches", "list"]:
            page = int(params.get("page", 1))
            chunk = self._match_list[(page - 1) * self._match_page:page * self._match_page]
            nxt = page + 1 if page * self._match_page < len(self._match_list) else None
            return {"results": chunk, "pagination": {"nextPage": nxt}}
        raise KeyError("offline Ticket has no endpoint %r" % (action,))


def install(ticket_cls, hyperparameter_cls=None, target_clip_cls=None):
    """Patch the reference's classes in place so broker.py / compute_matches.py see a drop-in."""
    if target_clip_cls is not None:                       # dynamic target adjustment: matrix formulas on the GPU
        from .target_clip import TargetClip as _TC
        for name in ("get_target_features", "avg_new_old_targets", "_previous", "dynamic_target_adjustment", "target_by_bagging",
                     "_draw", "_solve", "_stack"):
            setattr(target_clip_cls, name, getattr(_TC, name))
    for name in ("compute_similarities", "compute_scores", "lowest_scoring_user_match", "select_clips_to_review"):
        setattr(ticket_cls, name, getattr(TicketScoring, name))
    for name in ("feature_db", "feature_db_dtype", "device"):
        if not hasattr(ticket_cls, name):
            setattr(ticket_cls, name, getattr(TicketScoring, name))
    if hyperparameter_cls is not None:
        from .hyperparameter import Hyperparameter
        hyperparameter_cls.optimize_weights = Hyperparameter.optimize_weights
        hyperparameter_cls.fine_tune = Hyperparameter.fine_tune
        hyperparameter_cls._quad_fit = staticmethod(Hyperparameter._quad_fit)
