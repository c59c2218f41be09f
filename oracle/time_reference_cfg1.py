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
    out = {"clips": n, "compute_similarities_s": t_sim, "compute_scores_s": t_score, "select_clips_to_review_s": t_sel,
           "queries_per_s": 1.0 / (t_sim + t_score + t_sel), "records_build_s": t_records, "cpu_count": os.cpu_count(),
           "note": "reference Ticket methods, single-threaded by construction, feature lists as the REST API hands them over"}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
