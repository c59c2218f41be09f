"""Test program (GPU box): the broker's side of a served ShardedFeatureDB.  Started by tests/test_sharded_db_gpu.py as a process
of its own: ``ShardedFeatureDB.open`` starts the worker ranks (shard_worker.py) and joins them as rank 0 -- here every rank on the
box's ONE card over gloo (VQ_DIST_BACKEND=gloo) or ONE rank over RCCL -- and a query round runs through the Ticket seam on the
sharded database and on an ordinary one-GPU FeatureDB.  Prints ``ok``; any difference is an assertion."""
import os
import random
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)
os.environ.setdefault("COMPUTE_EPS", "0.000003")


def main():
    world, backend = int(sys.argv[1]), sys.argv[2]
    import video_query_algorithms_amd as vqa
    from video_query_algorithms_amd.feature_store import save_store
    from video_query_algorithms_amd.sharded_db import ShardedFeatureDB
    from _helpers import DEFAULT_WEIGHTS, SEED, STREAMS, golden_json, golden_npy, records_from_dense
    g = golden_json("synth_small.json")
    x = golden_npy("synth_small_x.npy")
    ids = np.asarray(g["clip_order"])
    recs = records_from_dense(x, ids, [1, 2, 3])
    with tempfile.TemporaryDirectory() as d:
        store = save_store(os.path.join(d, "store"), x, ids, STREAMS, [1, 2, 3])
        sdb = ShardedFeatureDB.open(store, gpus=[0] * world, backend=backend)      # before this process's first GPU call
        try:
            one = vqa.FeatureDB.from_store(store)
            ml = [{"video_clip": int(c), "user_match": v} for c, v in zip(ids[[3, 9, 20, 31, 40, 41, 50]], [True, True, False, True, None, False, True])]
            rounds = []
            for db in (one, sdb):
                tk = vqa.Ticket({"query_id": 1, "video_id": 1, "ref_clip": 0, "ref_clip_id": g["ref_clip_id"], "search_set": 1,
                                 "number_of_matches_to_review": 20, "user_matches": g["user_matches"]}, records=recs, feature_db=db)
                hp = vqa.Hyperparameter(DEFAULT_WEIGHTS, 0.8, g["ballast"], 0.35, 0.3, STREAMS, "global_pool", 1, 0.7, "bagging", 3)
                tk.target = vqa.TargetClip(tk, hp)
                tk.target.get_target_features()
                tk.compute_similarities(hp)
                tk.compute_scores(DEFAULT_WEIGHTS)
                random.seed(a=SEED)
                tk.select_clips_to_review(0.8, 20, 0.35)
                first = (tk._avg.copy(), tk._n_e.copy(), tk._score_values.copy(), list(tk.matches.items()), db.topk(10), db.min_score([5, 40, 63]))
                tk.matches = g["labelled"]
                hp.optimize_weights(tk)                                           # scores_grid over the labelled rows
                fit = (dict(hp.weights), hp.threshold)
                # a revise round with dynamic target adjustment: the validated clips are rows of the resident database
                tk2 = vqa.Ticket({"query_id": 1, "video_id": 1, "ref_clip": 0, "ref_clip_id": int(ids[7]), "search_set": 1,
                                  "dynamic_target_adjustment": True, "latest_query_result": {"id": 5, "round": 1, "bootstrapped_target": None},
                                  "match_list": ml, "match_page_size": 3, "user_matches": {}}, records=recs, feature_db=db)
                tk2.target = vqa.TargetClip(tk2, hp)
                random.seed(a=SEED)
                tk2.target.get_target_features()
                t_boot = np.array([[tk2.target.target_features[st][sp] for sp in (1, 2, 3)] for st in STREAMS])
                tk2.compute_similarities(hp)
                tk2.compute_scores(DEFAULT_WEIGHTS)
                rounds.append((first, fit, t_boot, tk2._score_values.copy(), random.random()))
            (f1, fit1, tb1, sc1, r1), (f2, fit2, tb2, sc2, r2) = rounds
            assert (f1[0] == f2[0]).all() and (f1[1] == f2[1]).all(), "similarities differ between one GPU and the shards"
            assert (f1[2] == f2[2]).all(), "scores differ"
            assert f1[3] == f2[3], "review sets differ"
            assert (f1[4][0] == f2[4][0]).all() and (f1[4][1] == f2[4][1]).all() and f1[5] == f2[5], "top-k / min differ"
            assert all(abs(s - w) <= 1e-12 for (_, s), (_, w) in zip(f2[3], g["select_default"]))
            assert [c for c, _ in f2[3]] == [c for c, _ in g["select_default"]]
            assert fit1 == fit2, (fit1, fit2)
            assert r1 == r2                                                       # the generator advanced identically
            assert np.abs(tb1 - tb2).max() <= 1e-12 * np.abs(tb1).max(), np.abs(tb1 - tb2).max()
            assert np.abs(sc1 - sc2).max() <= 1e-9
            # the 16-query pass, sharded: score slices gathered per query
            tb = np.stack([one.set_query_from_row(r) for r in (1, 7, 30)])
            wb = np.array([[1.0, 1.5], [1.0, 0.8], [1.0, 2.2]])
            assert (one.scan_batch(tb, wb) == sdb.scan_batch(tb, wb)).all()
            assert (sdb.set_query_from_row(41) == one.set_query_from_row(41)).all()
            # several tickets (and threads) on the ONE served database: every round gets what it gets alone, and the workers never
            # see one operation's header with another's payload (tests/_round_threads.py)
            from _round_threads import check_shared_database
            check_shared_database(vqa, sdb, recs, [int(c) for c in ids[[0, 7, 19, 33]]], g["labelled"], STREAMS, DEFAULT_WEIGHTS,
                                  threads=2, repeats=3)
            one.close()
        finally:
            sdb.close()
        assert all(p.returncode == 0 for p in sdb._workers), [p.returncode for p in sdb._workers]
    print("ok", flush=True)


if __name__ == "__main__":
    main()
