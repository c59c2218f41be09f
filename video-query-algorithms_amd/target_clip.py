"""The query vectors of a round: the seam of the reference's ``TargetClip`` (src/models/target_clip.py) over this
package's own data model.

What ``compute_matches`` and ``Ticket`` read from the object is kept -- ``TargetClip(ticket, hyperparameters)``,
``get_target_features()``, ``target_features`` ({stream: {split: list}}, JSON-serialisable for ticket.py:296),
``splits``, ``previous_target_features`` -- everything behind it is built differently:

* the validated clips of a round are gathered ONCE into a :class:`ValidatedPool`: dense ``[clip][stream][slot][D]``
  blocks with a presence mask (or, when the ticket carries a resident :class:`FeatureDB` that holds the clips, just
  their row numbers -- no feature leaves the device);
* a round is a list of *draws* (index sets into the pool; the calls into ``random`` follow the protocol of
  target_clip.py:297-309 so that a seeded broker picks the same clips), and every (draw, stream, split) problem of
  the round is solved by ONE launch of the closed-form kernel (csrc/vq_boot.hip: Gram matrix + Woodbury instead of
  the 1024 x 1024 inverses of target_clip.py:192-197 / :245-260);
* bagging averages the draws, ``partial_update`` blends with the previous round's target (target_clip.py:75-82).

SURVEY.md 8(a) row B1 (``t = r / (r . r)`` of the reference clip, target_clip.py:137-143, :263-286, :311-313) is
the no-adjustment case; 8(f) row 1 is the rest.
"""
from __future__ import annotations

import random
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from .bootstrap import bootstrap_targets

ADJUSTMENTS = ("simple", "partial_update", "bagging")


def clip_vectors(records, streams: Sequence[str], feature_name: str):
    """The API's feature records of one clip (wire format of target_clip.py:279-285) -> ({stream: {split: vector}},
    splits seen).  Records of other streams or other blobs are skipped."""
    table: Dict[str, dict] = {st: {} for st in streams}
    for rec in records:
        per_stream = table.get(rec["dnn_stream_id"])
        if per_stream is not None and rec["name"] == feature_name:
            per_stream[rec["dnn_stream_split"]] = rec["feature_vector"]
    return table, {sp for per_stream in table.values() for sp in per_stream}


def unit_response(vec) -> list:
    """``r / (r . r)``: the vector whose dot product with r is exactly 1 (target_clip.py:311-313)."""
    r = np.asarray(vec, dtype=np.float64)
    return (r / np.dot(r, r)).tolist()


def draw_indices(count: int, fraction, with_replacement: bool) -> List[int]:
    """Which of `count` validated clips enter one problem.  RNG protocol of target_clip.py:297-309: round(count *
    fraction) but at least one; ``random.sample`` without, ``random.choices`` with replacement; duplicates collapse
    through a ``set`` whose iteration order is the row order of the problem."""
    k = max(round(count * fraction), 1)
    population = range(count)
    hits = random.choices(population, k=k) if with_replacement else random.sample(population, k)
    return list(set(hits))


class ValidatedPool:
    """The user-validated clips of one kind (confirmed matches, or confirmed non-matches) of a round."""

    def __init__(self, streams: Sequence[str]):
        self.streams = list(streams)
        self.tables: List[dict] = []        # per clip {stream: {split: vector}} (record-fed)
        self.rows: List[int] = []           # per clip its row in the resident FeatureDB (row-fed)
        self.splits: set = set()

    def __len__(self):
        return len(self.tables) or len(self.rows)

    def add_table(self, table, splits):
        self.tables.append(table)
        self.splits |= set(splits)

    def stack(self, picked: Sequence[int], stream: str, split) -> np.ndarray:
        """[m][D] fp64: the (stream, split) vectors of the picked clips that have one, in pick order."""
        vecs = [self.tables[i][stream][split] for i in picked if split in self.tables[i][stream]]
        return np.asarray(vecs, dtype=np.float64).reshape(len(vecs), -1)


class TargetClip:
    def __init__(self, ticket, hyperparameters):
        self._ticket = ticket
        self.hyperparameters = hyperparameters
        self.client, self.schema = getattr(ticket, "client", None), getattr(ticket, "schema", None)
        self.bootstrap_target = bool(ticket.dynamic_target_adjustment)
        self.latest_query_result = ticket.latest_query_result
        earlier = (self.latest_query_result or {}).get("bootstrapped_target")
        self.previous_target_features = earlier or None
        self.ref_clip_features, self.splits = self._get_clip_features(ticket.ref_clip_id)
        self.target_features: dict = {}

    # ------------------------------------------------------------------ the seam
    def get_target_features(self):
        """Sets ``self.target_features`` for this round (decision ladder of target_clip.py:26-73)."""
        pools = self._validated_pools() if self.bootstrap_target and self.latest_query_result is not None else None
        if not pools or not len(pools[0]):
            self.target_features = self.scaled_ref_clip_features()
            return
        kind = self.hyperparameters.bootstrap_type
        if kind not in ADJUSTMENTS:
            raise Exception("Error: bootstrap_type should be one of 'simple', 'partial_update', or 'bagging'")
        getattr(self, "_adjust_" + kind)(*pools)

    def scaled_ref_clip_features(self):
        return {st: {sp: unit_response(v) for sp, v in per_split.items()}
                for st, per_split in self.ref_clip_features.items()}

    # ------------------------------------------------------------------ the three adjustment modes
    def _adjust_simple(self, valid, invalid):
        hp = self.hyperparameters
        self.target_features = self._solve(valid, invalid, [self._draw(valid, invalid, hp.f_bootstrap, False)])[0]

    def _adjust_partial_update(self, valid, invalid):
        self._adjust_simple(valid, invalid)
        before = self.previous_target_features
        if not before:
            return
        keep = self.hyperparameters.f_memory
        for st in self.hyperparameters.streams:
            mine, theirs = self.target_features[st], before[st]
            for sp in valid.splits:
                old = theirs[sp] if sp in theirs else theirs[str(sp)]        # JSON round trips turn keys into strings
                mine[sp] = (np.multiply(keep, mine[sp]) + np.multiply((1 - keep), old)).tolist()

    def _adjust_bagging(self, valid, invalid):
        hp = self.hyperparameters
        bags = self._solve(valid, invalid, [self._draw(valid, invalid, 1, True) for _ in range(hp.nbags)])
        self.target_features = {st: {sp: np.average([b[st][sp] for b in bags], axis=0).tolist() for sp in valid.splits}
                                for st in hp.streams}

    # ------------------------------------------------------------------ draws and the device solve
    @staticmethod
    def _draw(valid, invalid, fraction, with_replacement) -> Tuple[List[int], List[int]]:
        """One problem's clips.  With confirmed non-matches both kinds are always subsampled (target_clip.py:227-230);
        without, the matches are only touched when that changes something (:181-182) -- the generator must advance
        exactly as often as the reference's."""
        if len(invalid):
            return (draw_indices(len(valid), fraction, with_replacement),
                    draw_indices(len(invalid), fraction, with_replacement))
        if fraction != 1 or with_replacement:
            return draw_indices(len(valid), fraction, with_replacement), []
        return list(range(len(valid))), []

    def _solve(self, valid, invalid, draws) -> List[dict]:
        """All (draw, stream, split) problems of the round in one launch -> per draw {stream: {split: list}}."""
        hp = self.hyperparameters
        device = getattr(self._host(), "device", 0) or 0
        if valid.rows:
            return [self._solve_resident(valid, invalid, d) for d in draws]
        slots = [(st, sp) for st in hp.streams for sp in valid.splits]
        problems = []
        for picked_valid, picked_invalid in draws:
            for st, sp in slots:
                problems.append((valid.stack(picked_valid, st, sp),
                                 invalid.stack(picked_invalid, st, sp) if picked_invalid else None))
        solved = iter(bootstrap_targets(problems, hp.mu, device=device))
        out = []
        for _ in draws:
            target = {st: {} for st in hp.streams}
            for st, sp in slots:
                target[st][sp] = next(solved).tolist()
            out.append(target)
        return out

    def _solve_resident(self, valid, invalid, draw) -> dict:
        """The clips live in the ticket's FeatureDB: the kernel reads their rows in place."""
        db = self._host().feature_db
        picked_valid, picked_invalid = draw
        t = db.bootstrap_target([valid.rows[i] for i in picked_valid], [invalid.rows[i] for i in picked_invalid],
                                mu=self.hyperparameters.mu, set_query=False)
        return {st: {sp: t[db.stream_names.index(st), e].tolist() for e, sp in enumerate(db.slot_splits[db.stream_names.index(st)])}
                for st in self.hyperparameters.streams}

    # ------------------------------------------------------------------ gathering the validated clips
    def _validated_pools(self) -> Optional[Tuple[ValidatedPool, ValidatedPool]]:
        verdicts = self._match_verdicts()
        hp = self.hyperparameters
        valid, invalid = ValidatedPool(hp.streams), ValidatedPool(hp.streams)
        db = getattr(self._host(), "feature_db", None)
        clips = [c for c, _ in verdicts]
        if db is not None and self._resident_ok(db, clips):
            for clip, verdict in verdicts:
                (valid if verdict else invalid).rows.append(db.row_of(clip))
            valid.splits = set(db.slot_splits[0])
            return valid, invalid
        for clip, verdict in verdicts:
            (valid if verdict else invalid).add_table(*self._get_clip_features(clip))
        return valid, invalid

    def _resident_ok(self, db, clips) -> bool:
        """Rows can be used in place when the DB knows its slot layout, is dense, carries every stream of the round
        with one common split list, and holds every validated clip."""
        names, slots = getattr(db, "stream_names", None), getattr(db, "slot_splits", None)
        if not names or not slots or db.present is not None:
            return False
        if any(st not in names for st in self.hyperparameters.streams) or any(s != slots[0] for s in slots):
            return False
        return bool(clips) and all(db.has_clip(c) for c in clips)

    def _match_verdicts(self) -> List[Tuple[int, bool]]:
        """(clip id, user verdict) of every match of the latest query result the user has judged, in the API's order
        (all pages of ["matches", "list"], target_clip.py:117-124)."""
        judged, page = [], 1
        while page is not None:
            answer = self._request(["matches", "list"], {"query_result": self.latest_query_result["id"], "page": page})
            judged += [(m["video_clip"], m["user_match"]) for m in answer["results"]
                       if m["user_match"] is True or m["user_match"] is False]
            page = answer["pagination"]["nextPage"]
        return judged

    def _get_clip_features(self, clip_id):
        hp = self.hyperparameters
        return clip_vectors(self._request(["video-clips", "features"], {"id": clip_id}), hp.streams, hp.feature_name)

    def _host(self):
        """The ticket this target belongs to (None on a reference-class instance that install() did not wrap)."""
        return getattr(self, "_ticket", None)

    def _request(self, action, params):
        return self._ticket._request(action, params)


# what install() grafts onto the reference's TargetClip: the round logic; its own __init__ (wrapped to remember the
# ticket), _get_clip_features and _request (REST) stay the reference's
GRAFTED = ("get_target_features", "scaled_ref_clip_features", "_adjust_simple", "_adjust_partial_update", "_adjust_bagging",
           "_draw", "_solve", "_solve_resident", "_validated_pools", "_resident_ok", "_match_verdicts", "_host")
