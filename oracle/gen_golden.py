"""Generate golden vectors for hot path B by running the REFERENCE ITSELF.

Runs only in the build container (needs /root/reference); the vectors it writes
under tests/golden/ are what travels.  TEST INFRASTRUCTURE ONLY.

Recipe: register an empty ``coreapi`` stub, put
/root/reference/src on sys.path, import ``models`` unmodified, and drive
``Ticket`` / ``TargetClip`` / ``Hyperparameter`` through an in-memory
``_request``.  Nothing of the reference's source is copied; the outputs are data.

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden.py
"""
import hashlib
import json
import os
import random
import sys
import types
import warnings

import numpy as np

sys.dont_write_bytecode = True
REF = os.environ.get("VQ_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
sys.path.insert(0, HERE)

os.environ.setdefault("COMPUTE_EPS", "0.000003")   # README.md of the reference: COMPUTE_EPS example value
COMPUTE_EPS = float(os.environ["COMPUTE_EPS"])
SEED = "73459912436"                                # RANDOM_SEED example in the reference README

STREAMS = ("rgb", "warped_optical_flow")
DEFAULT_WEIGHTS = {"rgb": 1.0, "warped_optical_flow": 1.5}       # broker.py:36-39


def import_reference():
    stub = types.ModuleType("coreapi")
    stub.Client = object
    stub.auth = types.SimpleNamespace(TokenAuthentication=object)
    sys.modules["coreapi"] = stub
    sys.modules["coreapi.auth"] = stub.auth
    sys.path.insert(0, os.path.join(REF, "src"))
    import models  # noqa: F401  (the reference package)
    from models import Ticket, TargetClip, Hyperparameter
    return Ticket, TargetClip, Hyperparameter


Ticket, TargetClip, Hyperparameter = import_reference()


class FakeTicket(Ticket):
    """Ticket with the REST client replaced by in-memory record lists."""

    def __init__(self, records, ref_clip_id, user_matches=None, matches=None):
        self.client = None
        self.schema = None
        self.query_id = 1
        self.video_id = 1
        self.ref_clip = 0
        self.ref_clip_id = ref_clip_id
        self.search_set = 1
        self.number_of_matches_to_review = 20
        self.dynamic_target_adjustment = False
        self.latest_query_result = None
        self.matches = matches if matches is not None else []
        self.user_matches = user_matches or {}
        self.target = None
        self.similarities = {}
        self.scores = {}
        self._records = records

    def _request(self, action, params):
        if action == ["search-sets", "features"]:
            return self._records
        if action == ["video-clips", "features"]:
            return [r for r in self._records if r["video_clip_id"] == params["id"]]
        raise KeyError(action)


def make_target(ticket, hp):
    TargetClip._request = lambda self, action, params: ticket._request(action, params)
    tgt = TargetClip(ticket, hp)
    tgt.get_target_features()
    return tgt


def make_hp(ballast=0.0):
    # broker.py:36-59 defaults
    return Hyperparameter(DEFAULT_WEIGHTS, 0.8, ballast, 0.35, 0.0, STREAMS, "global_pool", 1, 0.7, "bagging", 3)


def records_from_dense(x, clip_ids, splits, present=None, order="split_major"):
    """[N,S,E,D] -> API records (Python lists of floats), in a chosen record order."""
    n, s, e, d = x.shape
    recs = []
    if order == "split_major":
        it = ((si, ei, ci) for ei in range(e) for si in range(s) for ci in range(n))
    else:  # clip_major
        it = ((si, ei, ci) for ci in range(n) for si in range(s) for ei in range(e))
    for si, ei, ci in it:
        if present is not None and not present[ci, si, ei]:
            continue
        recs.append({"dnn_stream_id": STREAMS[si], "dnn_stream_split": splits[ei], "name": "global_pool",
                     "video_clip_id": int(clip_ids[ci]), "feature_vector": x[ci, si, ei].astype(np.float64).tolist()})
    return recs


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def run_case(x, clip_ids, splits, ref_clip_id, present=None, order="split_major", user_matches=None,
             labelled=None, ballast=0.0, with_optimize=True, extra_records=()):
    """Drive the reference end to end for one data set; returns a dict of plain outputs."""
    recs = records_from_dense(x, clip_ids, splits, present, order) + list(extra_records)
    hp = make_hp(ballast)
    tk = FakeTicket(recs, ref_clip_id, user_matches=user_matches)
    tk.target = make_target(tk, hp)
    tk.compute_similarities(hp)
    out = {}
    out["target_splits"] = sorted(tk.target.splits)
    out["target"] = {st: {str(sp): v for sp, v in d.items()} for st, d in tk.target.target_features.items()}
    order_ids = list(tk.similarities.keys())
    out["clip_order"] = order_ids
    out["sim_avg"] = [[tk.similarities[c][st][0] if st in tk.similarities[c] else None for st in STREAMS]
                      for c in order_ids]
    out["sim_n"] = [[tk.similarities[c][st][1] if st in tk.similarities[c] else 0 for st in STREAMS]
                    for c in order_ids]
    tk.compute_scores(DEFAULT_WEIGHTS)
    out["scores_default"] = [float(tk.scores[c]) for c in order_ids]
    random.seed(a=SEED)
    tk.select_clips_to_review(0.8, 20, 0.35)
    out["select_default"] = [[int(k), float(v)] for k, v in tk.matches.items()]
    random.seed(a=SEED)
    tk.select_clips_to_review(0.8, 6, 0.5)
    out["select_max6"] = [[int(k), float(v)] for k, v in tk.matches.items()]
    low, low_clip = tk.lowest_scoring_user_match()
    out["lowest_user_match"] = [float(low), low_clip]
    # finalize path: compute_matches.py:79-89
    near = max(0.8 - low, 0) / max(1 - 0.8, COMPUTE_EPS)
    random.seed(a=SEED)
    tk.select_clips_to_review(0.8, float("inf"), near)
    out["finalize_near_miss"] = near
    out["select_finalize"] = [[int(k), float(v)] for k, v in tk.matches.items()]
    if with_optimize and labelled is not None:
        tk.matches = labelled
        hp.optimize_weights(tk)
        out["opt_weights"] = {k: float(v) for k, v in hp.weights.items()}
        out["opt_threshold"] = float(hp.threshold)
        out["scores_after_optimize"] = [float(tk.scores[c]) for c in order_ids]   # left at w = 2.45
        tk.compute_scores(hp.weights)
        out["scores_opt"] = [float(tk.scores[c]) for c in order_ids]
    return out


def synth_labels(scores, clip_ids, n_lab=20, seed=5):
    """Synthetic review round: top-scoring clips, labels loosely correlated with score."""
    rng = np.random.default_rng(seed)
    order = np.argsort(-np.asarray(scores), kind="stable")[:n_lab]
    labelled = []
    for rank, row in enumerate(order):
        r = rng.random()
        user = None if r < 0.2 else bool(rank < n_lab // 2) ^ bool(r > 0.85)
        labelled.append({"video_clip": int(clip_ids[row]), "user_match": user,
                         "is_match": bool(scores[row] >= 0.8)})
    return labelled


def rule_labels(sim_avg, clip_ids, w_star, th_star, n_lab=30):
    """Labels generated by a hidden (w*, th*) rule so that the loss minimum is interior to the grid."""
    from sim_oracle import dense_scores
    avg = np.array(sim_avg, dtype=np.float64)
    s_star = dense_scores(avg, [1.0, w_star])
    s_def = dense_scores(avg, [1.0, 1.5])
    order = np.argsort(-s_def, kind="stable")[:n_lab]
    labelled = []
    for rank, row in enumerate(order):
        user = None if rank % 7 == 6 else bool(s_star[row] >= th_star)
        labelled.append({"video_clip": int(clip_ids[row]), "user_match": user,
                         "is_match": bool(s_def[row] >= 0.8)})
    return labelled


def dump(name, obj):
    path = os.path.join(OUT, name)
    with open(path, "w") as f:
        json.dump(obj, f)
    print("wrote", path, os.path.getsize(path), "bytes")


def case_synth_small():
    """64 clips, dense, fp32-exact inputs regenerated from the seed by the tests."""
    from sim_oracle import cfg1_features
    n, e = 64, 3
    x = cfg1_features(n=n, e=e, seed=11)
    # plant structure: a few clips close to the reference clip so the match band is populated
    rng = np.random.default_rng(12)
    ref_row = 7
    plant = [(3, 0.97, 0.80), (9, 0.93, 0.95), (20, 0.9, 0.70), (21, 0.88, 0.92), (33, 0.84, 0.60), (40, 0.8, 0.9),
             (41, 0.78, 0.85), (50, 0.74, 0.93), (51, 0.72, 0.66), (60, 0.69, 0.88), (12, 0.95, 0.75),
             (13, 0.86, 0.97), (14, 0.82, 0.58), (15, 0.76, 0.9), (25, 0.9, 0.5), (26, 0.6, 0.96), (27, 0.88, 0.62),
             (28, 0.65, 0.9)]
    for row, a0, a1 in plant:       # different closeness per stream, so the stream weight matters
        for s, a in ((0, a0), (1, a1)):
            noise = np.abs(rng.standard_normal(x[row, s].shape).astype(np.float32)) \
                * x[ref_row, s].mean(axis=-1, keepdims=True)
            x[row, s] = (np.float32(a) * x[ref_row, s] + np.float32(1 - a) * noise).astype(np.float32)
    clip_ids = np.arange(1, n + 1) * 3 + 100
    splits = [1, 2, 3]
    user = {str(int(clip_ids[3])): True, str(int(clip_ids[20])): True, str(int(clip_ids[50])): True,
            str(int(clip_ids[5])): False}
    base = run_case(x, clip_ids, splits, int(clip_ids[ref_row]), user_matches=user, with_optimize=False)
    labelled = rule_labels(base["sim_avg"], base["clip_order"], w_star=1.4, th_star=0.85)
    out = run_case(x, clip_ids, splits, int(clip_ids[ref_row]), user_matches=user, labelled=labelled, ballast=0.3)
    # a second labelled round whose optimum sits on the grid border (no parabola refinement)
    lab2 = synth_labels(base["scores_default"], base["clip_order"])
    out_b = run_case(x, clip_ids, splits, int(clip_ids[ref_row]), user_matches=user, labelled=lab2, ballast=0.0)
    out["border"] = {"labelled": lab2, "ballast": 0.0, "opt_weights": out_b["opt_weights"],
                     "opt_threshold": out_b["opt_threshold"]}
    out["inputs_sha256"] = sha(x)
    out["labelled"] = labelled
    out["user_matches"] = user
    out["ref_clip_id"] = int(clip_ids[ref_row])
    out["ballast"] = 0.3
    np.save(os.path.join(OUT, "synth_small_x.npy"), x)          # 1.5 MB: committed, so no RNG dependence
    dump("synth_small.json", out)


def case_ragged():
    """12 clips with missing splits / shuffled record order: exercises n_e and first-seen order."""
    from sim_oracle import cfg1_features
    n, e = 12, 3
    x = cfg1_features(n=n, e=e, seed=21)
    x[4] = np.float32(0.9) * x[2] + np.float32(0.1) * x[4]
    x[9] = np.float32(0.8) * x[2] + np.float32(0.2) * x[9]
    clip_ids = np.array([50, 7, 19, 3, 88, 41, 42, 43, 5, 64, 65, 2])
    present = np.ones((n, 2, e), dtype=bool)
    present[1, :, 0] = False        # clip lacks split 1 entirely -> first seen under split 2
    present[5, 0, 2] = False        # rgb lacks split 3
    present[6, 1, 1] = False        # flow lacks split 2
    present[8, :, 1:] = False       # only split 1
    present[10, :, 0:2] = False     # only split 3 -> seen last
    splits = [1, 2, 3]
    # an irrelevant record (other feature name / unknown stream / unknown split) must be ignored
    extra = [{"dnn_stream_id": "rgb", "dnn_stream_split": 1, "name": "fc-action", "video_clip_id": 999,
              "feature_vector": [1.0] * 1024},
             {"dnn_stream_id": "audio", "dnn_stream_split": 1, "name": "global_pool", "video_clip_id": 998,
              "feature_vector": [1.0] * 1024},
             {"dnn_stream_id": "rgb", "dnn_stream_split": 9, "name": "global_pool", "video_clip_id": 997,
              "feature_vector": [1.0] * 1024}]
    out = run_case(x, clip_ids, splits, 19, present=present, order="clip_major", with_optimize=False,
                   extra_records=extra)
    out["present"] = present.astype(int).tolist()
    out["clip_ids"] = clip_ids.tolist()
    out["ref_clip_id"] = 19
    np.save(os.path.join(OUT, "ragged_x.npy"), x)
    dump("ragged.json", out)


def load_real_video(video_dir):
    """Parse the reference's shipped CSVs (api_load_records.py:41-58 rules) -> ids, x[N,S,3,1024] fp64."""
    xs = {}
    ids = None
    for ei, split in enumerate((1, 2, 3)):
        for si, st in enumerate(STREAMS):
            path = os.path.join(video_dir, "UCF101_split%d" % split, "%s_global_pool_features.csv" % st)
            with open(path) as f:
                f.readline()
                rows = [ln.rstrip("\n").split(",") for ln in f]
            cid = np.array([int(r[0]) for r in rows])
            vals = np.array([[float(v) for v in r[1:]] for r in rows], dtype=np.float64)
            if ids is None:
                ids = cid
            assert (ids == cid).all()
            xs[(si, ei)] = vals
    n = len(ids)
    x = np.empty((n, 2, 3, 1024), dtype=np.float64)
    for (si, ei), v in xs.items():
        x[:, si, ei] = v
    return ids, x


def case_real_subset():
    """24 real clips (fp64 values exactly as shipped) incl. ref clip 10 and the survey's top-10."""
    vdir = os.path.join(REF, "data", "features", "stock-video-clips_features", "DowntownBrooklynDrive_480p")
    ids, x = load_real_video(vdir)
    # whole video first: survey anchor: top-10 = [10,11,12,80,40,6,16,39,85,31]
    full = run_case(x, ids, [1, 2, 3], 10, with_optimize=False)
    sc = np.array(full["scores_default"])
    top10 = [full["clip_order"][i] for i in np.argsort(-sc, kind="stable")[:10]]
    print("full-video top10:", top10, "min/median/max", sc.min(), np.median(sc), sc.max())
    keep = sorted(set(top10) | {1, 2, 3, 5, 20, 25, 30, 45, 50, 55, 60, 70, 75, 87})
    rows = np.array([int(np.where(ids == k)[0][0]) for k in keep])
    xs = np.ascontiguousarray(x[rows])
    user = {"11": True, "12": True, "80": True, "20": False, "70": True}   # clip 70 scores < 0.8: near_miss > 0
    base = run_case(xs, ids[rows], [1, 2, 3], 10, user_matches=user, with_optimize=False)
    labelled = synth_labels(base["scores_default"], base["clip_order"], n_lab=12, seed=6)
    out = run_case(xs, ids[rows], [1, 2, 3], 10, user_matches=user, labelled=labelled, ballast=0.0)
    out["labelled"] = labelled
    out["user_matches"] = user
    out["ref_clip_id"] = 10
    out["ballast"] = 0.0
    out["clip_ids"] = ids[rows].tolist()
    out["full_video_top10"] = [int(t) for t in top10]
    out["full_video_scores_minmedmax"] = [float(sc.min()), float(np.median(sc)), float(sc.max())]
    np.save(os.path.join(OUT, "real_subset_x.npy"), xs)            # 24*6*1024 fp64 = 1.2 MB
    dump("real_subset.json", out)


def case_cfg1():
    """BASELINE config[0]: 10k x 1024, S=2, E=3 (SURVEY 8(d) cfg 1). Inputs regenerated from the seed."""
    from sim_oracle import cfg1_features
    n = 10000
    x = cfg1_features(n=n, e=3, seed=0)
    clip_ids = np.arange(1, n + 1)
    out = run_case(x, clip_ids, [1, 2, 3], 8, with_optimize=False)      # ref clip = row 7 (id = row+1)
    base_scores = out["scores_default"]
    labelled = synth_labels(base_scores, out["clip_order"])
    out2 = run_case(x, clip_ids, [1, 2, 3], 8, labelled=labelled)
    np.savez_compressed(os.path.join(OUT, "cfg1_10k.npz"),
                        sim_avg=np.array(out["sim_avg"], dtype=np.float64),
                        scores_default=np.array(base_scores, dtype=np.float64),
                        scores_opt=np.array(out2["scores_opt"], dtype=np.float64),
                        scores_after_optimize=np.array(out2["scores_after_optimize"], dtype=np.float64))
    meta = {"inputs_sha256": sha(x), "ref_clip_id": 8, "select_default": out["select_default"],
            "select_finalize_len": len(out["select_finalize"]), "labelled": labelled,
            "opt_weights": out2["opt_weights"], "opt_threshold": out2["opt_threshold"],
            "target_sha256": sha(np.array([[out["target"][st][str(sp)] for sp in (1, 2, 3)] for st in STREAMS]))}
    dump("cfg1_10k.json", meta)


if __name__ == "__main__":
    warnings.simplefilter("ignore", DeprecationWarning)      # random.sample(dict view) on 3.10
    os.makedirs(OUT, exist_ok=True)
    which = sys.argv[1:] or ["synth_small", "ragged", "real_subset", "cfg1"]
    for w in which:
        globals()["case_" + w]()
    meta = {"python": sys.version, "numpy": np.__version__, "COMPUTE_EPS": COMPUTE_EPS, "RANDOM_SEED": SEED,
            "reference": "PARC-projects/video-query-algorithms @ /root/reference (imported unmodified, coreapi stubbed)"}
    dump("META.json", meta)
