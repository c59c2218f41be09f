"""Test infrastructure: query rounds of SEVERAL tickets on ONE resident database.

The query, the averaged similarities and the scores are state of the database handle, and a round is several calls
(compute_similarities -> optimize_weights -> compute_scores -> select_clips_to_review, compute_matches.py:58-89).  Whatever the
interleaving, every round must produce what it produces alone."""
import sys
import threading

import numpy as np


def one_round(vqa, db, recs, ref_clip_id, labelled, streams, default_weights, pause=None):
    tk = vqa.Ticket({"query_id": 1, "video_id": 1, "ref_clip": 0, "ref_clip_id": int(ref_clip_id), "search_set": 1,
                     "number_of_matches_to_review": 20, "dynamic_target_adjustment": False, "user_matches": {}}, records=recs, feature_db=db)
    hp = vqa.Hyperparameter(default_weights, 0.8, 0.1, 0.35, 0.0, streams, "global_pool", 1, 0.7, "bagging", 3)
    tk.target = vqa.TargetClip(tk, hp)
    tk.target.get_target_features()
    tk.compute_similarities(hp)
    if pause:
        pause("similarities")
    tk.matches = labelled
    hp.optimize_weights(tk)
    if pause:
        pause("weights")
    tk.compute_scores(hp.weights)
    if pause:
        pause("scores")
    tk.select_clips_to_review(hp.threshold, float("inf"), 0.35)      # everything above the lower limit: no random draw decides the SET
    return {"avg": tk._avg.copy(), "n_e": tk._n_e.copy(), "scores": tk._score_values.copy(), "weights": dict(hp.weights),
            "threshold": hp.threshold, "matches": dict(tk.matches)}


def same(a, b):
    return ((a["avg"] == b["avg"]).all() and (a["n_e"] == b["n_e"]).all() and (a["scores"] == b["scores"]).all() and a["weights"] == b["weights"]
            and a["threshold"] == b["threshold"] and a["matches"] == b["matches"])


def check_shared_database(vqa, db, recs, ref_clips, labelled, streams, default_weights, threads=3, repeats=6):
    """(1) every ticket alone; (2) a deterministic worst case: ticket B runs a whole round inside every gap of ticket A's round;
    (3) `threads` threads running rounds with different targets at once."""
    alone = {c: one_round(vqa, db, recs, c, labelled, streams, default_weights) for c in ref_clips}
    assert not same(alone[ref_clips[0]], alone[ref_clips[1]])
    a, b = ref_clips[0], ref_clips[1]
    seen = []

    def intruder(where):
        seen.append(where)
        assert same(one_round(vqa, db, recs, b, labelled, streams, default_weights), alone[b]), ("intruder after", where)
    assert same(one_round(vqa, db, recs, a, labelled, streams, default_weights, pause=intruder), alone[a]), "a round interleaved with another"
    assert seen == ["similarities", "weights", "scores"]
    errors = []

    def worker(k):
        try:
            for r in range(repeats):
                c = ref_clips[(k + r) % len(ref_clips)]
                if not same(one_round(vqa, db, recs, c, labelled, streams, default_weights), alone[c]):
                    errors.append((k, r, c))
        except Exception as e:                              # noqa: BLE001 -- reported below
            errors.append((k, repr(e)))
    old = sys.getswitchinterval()
    sys.setswitchinterval(1e-6)                             # hand the interpreter over as often as possible
    try:
        ts = [threading.Thread(target=worker, args=(k,)) for k in range(threads)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
    finally:
        sys.setswitchinterval(old)
    assert not errors, errors
