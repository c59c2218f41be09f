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

import contextlib
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


def _db_lock(db):
    """The database's re-entrant lock (FeatureDB / ShardedFeatureDB); a database object without one gets no locking."""
    return getattr(db, "lock", None) or contextlib.nullcontext()


class TicketScoring:
    """Hot-path methods of the reference's Ticket, on the GPU.  Attributes read: ``target``
    (``target_features``, ``splits``), ``search_set``, ``ref_clip_id``, ``user_matches``; attributes
    written: ``similarities``, ``scores``, ``matches`` -- as in the reference."""

    feature_db: FeatureDB | None = None      # a resident DB may be attached up front
    _ahead = None                            # scores / review partition this round's compute_similarities or compute_scores already brought back
    _hp = None                               # the hyperparameters object of the latest compute_similarities
    _stream_gap = None                       # per stream: does any clip lack it (compute_scores' KeyError of ticket.py:177)
    _round = None                            # token of this ticket's latest compute_similarities (see _own_similarities)
    _score_weights = None
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
        # The target's vectors are Python lists (JSON-serialisable, ticket.py:296): turning 6 x 1 024 floats into an array is ~0.15 ms
        # of interpreter time, more than the whole device side of a 10k-clip round.  The array is kept on the target object and
        # reused while every vector IS the list object it was made from (TargetClip replaces a vector by a new list whenever it
        # changes it -- target_clip.py never assigns into one).
        refs = tuple(target_features[st][sp] for s, st in enumerate(stream_names) for sp in slot_splits[s]
                     if st in target_features and sp in target_features[st])
        cached = getattr(self.target, "_vq_query_array", None)
        if cached is not None and cached[0] == (db.S, db.E, db.D) and len(cached[1]) == len(refs) and all(a is b for a, b in zip(cached[1], refs)):
            t, slot_used = cached[2], cached[3]
        else:
            t = np.zeros((db.S, db.E, db.D), dtype=np.float64)
            slot_used = np.zeros((db.S, db.E), dtype=bool)
            for s, st in enumerate(stream_names):
                for e, sp in enumerate(slot_splits[s]):
                    if st in target_features and sp in target_features[st]:
                        t[s, e] = np.asarray(target_features[st][sp], dtype=np.float64)
                        slot_used[s, e] = True
            try:
                self.target._vq_query_array = ((db.S, db.E, db.D), refs, t, slot_used)
            except AttributeError:
                pass                                          # a target object that takes no attributes
        self._ahead = None
        with _db_lock(db):                    # one resident database may serve several tickets: these calls are one step
            db.restrict_slots(slot_used)      # per query; the database's own mask is never modified (no device call unless it changes)
            self._round = object()            # whose similarities the database holds now (see _own_similarities)
            ahead = self._look_ahead(hyperparameters, stream_names) if hasattr(db, "query_round") else None
            if ahead is not None:
                # The round in ONE call (vq_db_query_round): the reference calls compute_scores and select_clips_to_review right behind
                # this method (compute_matches.py:58-89) with the weights / threshold its hyperparameters object holds now, so scores and
                # review partition are computed and brought back with the similarities; compute_scores / select_clips_to_review use
                # them when they are called with exactly those arguments and run their own calls otherwise.
                w, th, lower = ahead
                r = db.query_round(t, weights=w, select=(th, lower))
                avg, n_e = r.avg, r.n_e
                self._ahead = {"w": w.tobytes(), "scores": r.scores, "select": (th, lower), "match_rows": r.match_rows,
                               "near_rows": r.near_rows, "near_argmax": r.near_argmax}
                db.sims_owner, db.scores_owner = self._round, (self._round, w.tobytes())
            else:
                db.set_query(t)
                db.scan(weights=None)
                avg, n_e = db.similarities()
                db.sims_owner, db.scores_owner = self._round, None
        self._hp = hyperparameters
        self._stream_names = stream_names
        self._avg, self._n_e = avg, n_e
        self._stream_gap = None
        self.similarities = SimilarityMap(db.clip_ids, stream_names, avg, n_e, db.row_of)

    @staticmethod
    def _review_band(hyperparameters):
        """(threshold, lower limit) select_clips_to_review will most likely be called with (compute_matches.py:78-88)."""
        th = getattr(hyperparameters, "threshold", None)
        if th is None:
            th = getattr(hyperparameters, "default_threshold", None)
        near = getattr(hyperparameters, "near_miss_default", None)
        if th is None or near is None:
            return None
        return float(th), float(th) - float(near) * (1 - float(th))          # the expression of ticket.py:316

    def _look_ahead(self, hyperparameters, stream_names):
        """The arguments compute_scores / select_clips_to_review are about to get if the caller is the reference's compute_matches:
        (weights [S], threshold, lower limit), or None when the hyperparameters object does not say."""
        weights = getattr(hyperparameters, "weights", None) or getattr(hyperparameters, "default_weights", None)
        band = self._review_band(hyperparameters)
        if not isinstance(weights, Mapping) or band is None:
            return None
        w = np.zeros(len(stream_names), dtype=np.float64)
        for stream_type, ws in weights.items():
            if stream_type not in stream_names:
                return None
            w[stream_names.index(stream_type)] = ws
        return w, band[0], band[1]

    # -- ticket.py:165-180 ------------------------------------------------------------------
    def compute_scores(self, weights):
        db = self.feature_db
        w = np.zeros(db.S, dtype=np.float64)
        if self._stream_gap is None:                 # stream -> does some clip have no similarity for it: one pass per round and stream
            self._stream_gap = {}
        for stream_type, ws in weights.items():
            s = self._stream_names.index(stream_type) if stream_type in self._stream_names else -1
            if s >= 0 and s not in self._stream_gap:
                self._stream_gap[s] = bool((self._n_e[:, s] == 0).any())
            if s < 0 or self._stream_gap[s]:
                raise KeyError(stream_type)          # vsim[stream_type] at ticket.py:177
            w[s] = ws
        ahead = getattr(self, "_ahead", None)
        if ahead is not None and ahead["w"] == w.tobytes():
            # computed from THIS round's similarities under exactly these weights and already on the host: nothing to ask the device
            # (whose scores may meanwhile be another ticket's: select_clips_to_review / _own_scores look after that)
            self._score_values = ahead["scores"]
            self._score_weights = w
            self.scores = ScoreMap(db.clip_ids, self._score_values, db.row_of)
            return
        with _db_lock(db):
            self._own_similarities()
            band = self._review_band(getattr(self, "_hp", None)) if hasattr(db, "query_round") else None
            if band is not None:                               # re-weighting + the partition select_clips_to_review is about to ask for
                r = db.query_round(None, weights=w, select=band)
                self._score_values = r.scores
                self._ahead = {"w": w.tobytes(), "scores": r.scores, "select": band, "match_rows": r.match_rows,
                               "near_rows": r.near_rows, "near_argmax": r.near_argmax}
            else:
                db.rescore(w)
                self._score_values = db.scores()
                self._ahead = None
            self._score_weights = w
            db.scores_owner = (self._round, w.tobytes())
        self.scores = ScoreMap(db.clip_ids, self._score_values, db.row_of)

    # -- a resident database shared between tickets -------------------------------------------------
    def _own_similarities(self):
        """The query, the averaged similarities and the scores are state of the database HANDLE, and a round is several calls
        (compute_similarities, optimize_weights, compute_scores, select_clips_to_review -- compute_matches.py:58-89).  When another
        ticket used the same resident database in between, this ticket's averaged similarities (kept on the host since
        compute_similarities) go back to the device before anything is computed from them: N x S x 12 bytes, only ever on a
        collision.  Call with the database's lock held."""
        db = self.feature_db
        if getattr(db, "sims_owner", self._round) is not self._round:
            db.write_avg(self._avg, self._n_e)
            db.sims_owner, db.scores_owner = self._round, None

    def _own_scores(self):
        """As above for the scores the selection kernels read: recomputed from this ticket's similarities and its last weights
        when the device holds somebody else's."""
        db = self.feature_db
        mine = (self._round, self._score_weights.tobytes())
        if getattr(db, "scores_owner", mine) != mine:
            self._own_similarities()
            db.rescore(self._score_weights)
            db.scores_owner = mine

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
        ahead = getattr(self, "_ahead", None)
        if ahead is not None and ahead["select"] == (threshold, lower_limit) and ahead["w"] == self._score_weights.tobytes():
            # the partition of exactly these scores under exactly this band came back with them
            match_rows, near_rows, near_argmax = ahead["match_rows"], ahead["near_rows"], ahead["near_argmax"]
        else:
            with _db_lock(db):
                self._own_scores()
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
        # clips the user has already judged stay in the review set whatever their score: the reference clip (when it is
        # part of the search set) and every confirmed match (ticket.py:346-356); a confirmed match that is not in the
        # search set is a KeyError there and here
        forced = [self.ref_clip_id] if self.ref_clip_id in self.scores else []
        forced += [int(clip) for clip, verdict in (self.user_matches or {}).items() if verdict is True]
        for clip in forced:
            matches[clip] = self.scores[clip]
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

    # -- the REST side of a query round, kept in memory (the reference posts these to the API: ticket.py:59-118,
    #    :276-299).  Enough for ``compute_matches(query_updates, hyperparameters)`` to run with no server; what the
    #    server would have received can be read back from ``ledger``.
    @property
    def ledger(self):
        return self.__dict__.setdefault("_ledger", {"process_state": [], "notes": [], "query_results": [], "matches": [],
                                                    "final_report": None})

    def change_process_state(self, process_state, message=None):
        self.ledger["process_state"].append(process_state)
        if message:
            self.add_note(message)
        return process_state

    def add_note(self, note):
        self.ledger["notes"].append(note)

    def catch_errors(self, job_type):
        """(fatal message, warning) -- the three consistency checks of ticket.py:80-110; the third one switches
        dynamic target adjustment off for the round when the user confirmed nothing."""
        fatal, warnings = [], []
        if self.ref_clip_id is None:
            fatal.append("*** Fatal Error: A video clip corresponding to the reference time does not exist in the database. ***")
        if job_type != "new" and not self.matches:
            fatal.append("*** Fatal Error: This is not a new query but there are 0 matches computed for the previous round.")
        if job_type != "new" and self.dynamic_target_adjustment is True \
                and not any(m["user_match"] is True for m in self.matches):
            warnings.append("*** Error: Dynamic target adjustment is True but there are no user matches provided for the "
                            "previous round. Changing dynamic target adjustment to False")
            self.dynamic_target_adjustment = False
        return "\n".join(fatal), "\n".join(warnings)

    def create_query_result(self, nround, hyperparameters):
        import json
        results = self.ledger["query_results"]
        results.append({"id": len(results) + 1, "round": nround, "match_criterion": hyperparameters.threshold,
                        "weights": [hyperparameters.weights[st] for st in hyperparameters.streams], "query": self.query_id,
                        "bootstrapped_target": json.dumps(self.target.target_features)})
        return results[-1]["id"]

    def add_matches_to_database(self, new_result_id):
        for clip, score in self.matches.items():
            self.ledger["matches"].append({"query_result": new_result_id, "score": score, "video_clip": clip,
                                           "user_match": self.user_matches.get(str(clip))})

    def create_final_report(self, hyperparameters, query_result_id):
        """The rows of the report of ticket.py:244-268: every selected clip, best score first (stable)."""
        criterion = self.ledger["query_results"][query_result_id - 1]["match_criterion"]

        def kind(clip, score):
            verdict = self.user_matches.get(str(clip))
            if verdict is not None:
                return "user-identified match" if verdict is True else "user-identified non-match"
            return "inferred match" if score >= criterion else "inferred non-match"
        rows = [[clip, kind(clip, score), score] for clip, score in self.matches.items()]
        rows.sort(key=lambda row: row[2], reverse=True)
        self.ledger["final_report"] = rows


def install(ticket_cls, hyperparameter_cls=None, target_clip_cls=None):
    """Patch the reference's classes in place so broker.py / compute_matches.py see a drop-in."""
    if target_clip_cls is not None:                       # dynamic target adjustment: matrix formulas on the GPU
        from . import target_clip as _tc
        for name in _tc.GRAFTED:
            setattr(target_clip_cls, name, _tc.TargetClip.__dict__[name])      # __dict__: staticmethods stay static
        construct = target_clip_cls.__init__

        def remember_ticket(self, ticket, hyperparameters):
            construct(self, ticket, hyperparameters)
            self._ticket = ticket                          # lets the round use rows of a resident ticket.feature_db
        target_clip_cls.__init__ = remember_ticket
    for name in ("compute_similarities", "compute_scores", "lowest_scoring_user_match", "select_clips_to_review",
                 "_own_similarities", "_own_scores", "_look_ahead"):
        setattr(ticket_cls, name, getattr(TicketScoring, name))
    ticket_cls._review_band = staticmethod(TicketScoring._review_band)
    for name in ("feature_db", "feature_db_dtype", "device", "_round", "_score_weights", "_ahead", "_hp", "_stream_gap"):
        if not hasattr(ticket_cls, name):
            setattr(ticket_cls, name, getattr(TicketScoring, name))
    if hyperparameter_cls is not None:
        from .hyperparameter import Hyperparameter
        for name in ("optimize_weights", "_labelled_rows", "_refine", "_pick"):
            setattr(hyperparameter_cls, name, Hyperparameter.__dict__[name])
