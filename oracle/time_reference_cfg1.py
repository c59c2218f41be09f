"""Time the UNMODIFIED reference on BASELINE configs[0] (SURVEY.md 8(d) cfg 1: 10 000 clips x 2 streams x 3 splits x
1024, ref clip = row 7) in the build container -- the reference cannot travel to the GPU box.  TEST INFRASTRUCTURE ONLY.

    PYTHONDONTWRITEBYTECODE=1 python oracle/time_reference_cfg1.py
"""
import json
import os
import random
import sys
import time

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as gg  # noqa: E402  (imports the reference package)
from sim_oracle import cfg1_features  # noqa: E402


def main():
    n = 10000
    x = cfg1_features(n=n, e=3, seed=0)
    ids = list(range(1, n + 1))
    t0 = time.perf_counter()
    recs = gg.records_from_dense(x, ids, [1, 2, 3])
    t_records = time.perf_counter() - t0
    hp = gg.make_hp()
    tk = gg.FakeTicket(recs, 8)
    tk.target = gg.make_target(tk, hp)
    t0 = time.perf_counter()
    tk.compute_similarities(hp)
    t_sim = time.perf_counter() - t0
    t0 = time.perf_counter()
    tk.compute_scores(gg.DEFAULT_WEIGHTS)
    t_score = time.perf_counter() - t0
    random.seed(a=gg.SEED)
    t0 = time.perf_counter()
    tk.select_clips_to_review(0.8, 20, 0.35)
    t_sel = time.perf_counter() - t0
    # BASELINE configs[4]: one weight-update round on the same resident search set -- the reference's own Hyperparameter.optimize_weights
    # (hyperparameter.py:29-76: 40 full rescoring passes) on 20 labels taken from the top of the current ranking, then compute_scores and
    # select_clips_to_review; compute_similarities of the round is the figure above (the reference recomputes it every round)
    ranked = sorted(tk.scores.items(), key=lambda kv: kv[1], reverse=True)[:20]
    tk.matches = [{"video_clip": c, "user_match": bool(i % 3 != 2), "is_match": bool(v >= 0.8)} for i, (c, v) in enumerate(ranked)]
    t0 = time.perf_counter()
    hp.optimize_weights(tk)
    t_opt = time.perf_counter() - t0
    t0 = time.perf_counter()
    tk.compute_scores(hp.weights)
    random.seed(a=gg.SEED)
    tk.select_clips_to_review(hp.threshold, 20, 0.35)
    t_after = time.perf_counter() - t0
    out = {"clips": n, "compute_similarities_s": t_sim, "compute_scores_s": t_score, "select_clips_to_review_s": t_sel,
           "queries_per_s": 1.0 / (t_sim + t_score + t_sel), "records_build_s": t_records, "cpu_count": os.cpu_count(),
           "optimize_weights_s": t_opt, "scores_and_select_after_update_s": t_after,
           "rounds_per_s": 1.0 / (t_sim + t_opt + t_after), "date": time.strftime("%Y-%m-%d"),
           "note": "reference Ticket / Hyperparameter methods imported unmodified, single-threaded by construction, feature lists as the REST API "
                   "hands them over; rounds_per_s = one weight-update round of BASELINE configs[4] (similarities + optimize_weights on 20 labels + "
                   "scores + select)"}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
