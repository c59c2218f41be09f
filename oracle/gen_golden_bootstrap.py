"""Golden vectors for target bootstrapping, produced by running the REFERENCE's own
``TargetClip`` (src/models/target_clip.py:26-261) in the build container.  TEST INFRASTRUCTURE ONLY.

Same recipe as oracle/gen_golden.py (coreapi stubbed, in-memory ``_request``); inputs are the committed
tests/golden/real_subset_x.npy (24 real clips, fp64) and synth_small_x.npy (64 clips, fp32).

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_bootstrap.py
"""
import json
import os
import random
import sys

import numpy as np

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as gg  # noqa: E402  (imports the reference package)

OUT = gg.OUT


class BootTicket(gg.FakeTicket):
    """Ticket of a query in round >= 2 with dynamic target adjustment on (ticket.py:38-57 attributes)."""

    def __init__(self, records, ref_clip_id, matches, previous_target=None, page_size=4):
        super().__init__(records, ref_clip_id)
        self.dynamic_target_adjustment = True
        self.latest_query_result = {"id": 5, "round": 1, "bootstrapped_target": previous_target}
        self._match_list = matches
        self._page = page_size

    def _request(self, action, params):
        if action == ["matches", "list"]:             # paginated, target_clip.py:114-121
            assert params["query_result"] == 5
            p = params["page"]
            chunk = self._match_list[(p - 1) * self._page:p * self._page]
            nxt = p + 1 if p * self._page < len(self._match_list) else None
            return {"results": chunk, "pagination": {"nextPage": nxt}}
        return super()._request(action, params)


def hp_for(bootstrap_type, mu, f_bootstrap, f_memory=0.7, nbags=3):
    return gg.Hyperparameter(gg.DEFAULT_WEIGHTS, 0.8, 0.0, 0.35, mu, gg.STREAMS, "global_pool", f_bootstrap, f_memory,
                             bootstrap_type, nbags)


def run(records, ref_clip_id, matches, bootstrap_type, mu, f_bootstrap, previous=None):
    hp = hp_for(bootstrap_type, mu, f_bootstrap)
    tk = BootTicket(records, ref_clip_id, matches, previous_target=previous)
    gg.TargetClip._request = lambda self, action, params: tk._request(action, params)
    tgt = gg.TargetClip(tk, hp)
    random.seed(a=gg.SEED)
    tgt.get_target_features()
    tf = tgt.target_features
    target = {st: {str(sp): [float(v) for v in np.asarray(vec).ravel()] for sp, vec in d.items()} for st, d in tf.items()}
    # downstream: similarities + default scores with this target (ticket.py:120-180)
    tk.target = tgt
    tk.compute_similarities(hp)
    tk.compute_scores(gg.DEFAULT_WEIGHTS)
    order = list(tk.similarities.keys())
    return {"bootstrap_type": bootstrap_type, "mu": mu, "f_bootstrap": f_bootstrap, "f_memory": 0.7, "nbags": 3,
            "matches": matches, "had_previous": previous is not None, "target": target,
            "clip_order": order, "scores_default": [float(tk.scores[c]) for c in order]}


def main():
    x = np.load(os.path.join(OUT, "real_subset_x.npy"))
    meta = json.load(open(os.path.join(OUT, "real_subset.json")))
    ids = np.array(meta["clip_ids"])
    recs = gg.records_from_dense(x, ids, [1, 2, 3])
    valid = [11, 12, 80, 40, 6, 16]
    invalid = [20, 70, 1, 55]
    unrated = [39, 85]
    def mk(v, iv):
        ms = [{"video_clip": int(c), "user_match": True} for c in v] + [{"video_clip": int(c), "user_match": False} for c in iv] \
            + [{"video_clip": int(c), "user_match": None} for c in unrated]
        random.Random(3).shuffle(ms)
        return ms
    cases = {}
    prev = run(recs, 10, mk(valid, []), "simple", 0.0, 1)["target"]       # a previous bootstrapped target
    prev_native = {st: {int(sp): v for sp, v in d.items()} for st, d in prev.items()}
    for name, v, iv, bt, mu, fb, pv in [
            ("simple_valid", valid, [], "simple", 0.0, 1, None),
            ("simple_valid_half", valid, [], "simple", 0.3, 0.5, None),
            ("simple_both_mu0", valid, invalid, "simple", 0.0, 1, None),
            ("simple_both_mu03", valid, invalid, "simple", 0.3, 1, None),
            ("simple_both_mu03_half", valid, invalid, "simple", 0.3, 0.5, None),
            ("partial_both", valid, invalid, "partial_update", 0.3, 1, prev_native),
            ("partial_noprev", valid, invalid, "partial_update", 0.3, 1, None),
            ("bagging_valid", valid, [], "bagging", 0.0, 1, None),
            ("bagging_both_mu0", valid, invalid, "bagging", 0.0, 1, None),
            ("bagging_both_mu03", valid, invalid, "bagging", 0.3, 1, None),
            ("no_valid_matches", [], invalid, "bagging", 0.3, 1, None),          # case 2: falls back to the ref clip
    ]:
        cases[name] = run(recs, 10, mk(v, iv), bt, mu, fb, pv)
        t = np.array(cases[name]["target"]["rgb"]["1"])
        print(name, "rgb/1 |t| max %.4e" % np.abs(t).max(), "top score ids",
              [cases[name]["clip_order"][i] for i in np.argsort(-np.array(cases[name]["scores_default"]), kind="stable")[:5]])
    cases["previous_target"] = prev
    # fp32 synthetic inputs (cfg-1 style), more rows
    xs = np.load(os.path.join(OUT, "synth_small_x.npy"))
    sids = np.arange(1, xs.shape[0] + 1) * 3 + 100
    srecs = gg.records_from_dense(xs, sids, [1, 2, 3])
    sv = [int(sids[i]) for i in (3, 9, 20, 21, 33, 40, 12, 13, 25, 27)]
    siv = [int(sids[i]) for i in (5, 30, 45, 55, 58, 61, 62)]
    ms = [{"video_clip": c, "user_match": True} for c in sv] + [{"video_clip": c, "user_match": False} for c in siv]
    cases["synth_bagging_mu03"] = run(srecs, int(sids[7]), ms, "bagging", 0.3, 1)
    cases["synth_simple_mu03"] = run(srecs, int(sids[7]), ms, "simple", 0.3, 1)
    # targets as one fp64 array per case ([stream][split][1024]); the rest as JSON
    arrays = {}
    for name, c in cases.items():
        tgt = c if name == "previous_target" else c.pop("target")
        arrays[name] = np.array([[tgt[st][str(sp)] for sp in (1, 2, 3)] for st in gg.STREAMS], dtype=np.float64)
    del cases["previous_target"]
    np.savez_compressed(os.path.join(OUT, "bootstrap_targets.npz"), **arrays)
    path = os.path.join(OUT, "bootstrap.json")
    with open(path, "w") as f:
        json.dump(cases, f)
    print("wrote", path, os.path.getsize(path), "bytes;", os.path.getsize(os.path.join(OUT, "bootstrap_targets.npz")), "bytes of targets")


if __name__ == "__main__":
    main()
