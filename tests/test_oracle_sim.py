"""CPU: the oracle (oracle/sim_oracle.py) against the golden vectors recorded from the reference itself
(oracle/gen_golden.py).  This is what "pins" the oracle; the GPU tests then compare the HIP path with it."""
import hashlib
import random

import numpy as np
import pytest

import sim_oracle as so
from _helpers import (DEFAULT_WEIGHTS, RAGGED_EXTRA, SEED, STREAMS, golden_json, golden_npy, golden_target_array,
                      records_from_dense)

CASES = [("synth_small", "split_major", None), ("ragged", "clip_major", RAGGED_EXTRA), ("real_subset", "split_major", None)]


def _load(name):
    g = golden_json(name + ".json")
    x = golden_npy(name + "_x.npy")
    ids = g.get("clip_ids") or g["clip_order"]
    present = np.array(g["present"], dtype=bool) if "present" in g else None
    return g, x, np.asarray(ids), present


def _faithful(name, order, extra):
    g, x, ids, present = _load(name)
    recs = records_from_dense(x, ids, [1, 2, 3], present, order, extra or ())
    ref_feats, splits = so.faithful_clip_features([r for r in recs if r["video_clip_id"] == g["ref_clip_id"]],
                                                  STREAMS, "global_pool")
    target = so.faithful_scaled_ref_clip_features(ref_feats)
    cand = so.faithful_candidate_features(recs, splits, STREAMS, "global_pool")
    sims = so.faithful_similarities(target, cand)
    return g, target, sims


@pytest.mark.parametrize("name,order,extra", CASES)
def test_faithful_restatement_matches_reference(name, order, extra):
    g, target, sims = _faithful(name, order, extra)
    assert list(sims.keys()) == g["clip_order"]                        # first-seen order
    for st in STREAMS:
        for sp, v in target[st].items():
            assert v == g["target"][st][str(sp)]                       # bit-exact: same numpy ops
    for c, avg_row, n_row in zip(g["clip_order"], g["sim_avg"], g["sim_n"]):
        for si, st in enumerate(STREAMS):
            if n_row[si]:
                assert sims[c][st][0] == avg_row[si] and sims[c][st][1] == n_row[si]
            else:
                assert st not in sims[c]
    scores = so.faithful_scores(sims, DEFAULT_WEIGHTS)
    assert [float(scores[c]) for c in g["clip_order"]] == g["scores_default"]
    um = g.get("user_matches", {})
    for key, mx, nm in (("select_default", 20, 0.35), ("select_max6", 6, 0.5)):
        random.seed(a=SEED)
        m = so.faithful_select(scores, g["ref_clip_id"], um, 0.8, mx, nm)
        assert [[int(k), float(v)] for k, v in m.items()] == g[key]
    low, clip = so.faithful_lowest_scoring_user_match(scores, um)
    assert [float(low), clip] == g["lowest_user_match"]
    near = so.finalize_near_miss(0.8, low, 0.000003)
    assert near == g["finalize_near_miss"]
    random.seed(a=SEED)
    m = so.faithful_select(scores, g["ref_clip_id"], um, 0.8, float("inf"), near)
    assert [[int(k), float(v)] for k, v in m.items()] == g["select_finalize"]


@pytest.mark.parametrize("name", ["synth_small", "real_subset"])
def test_faithful_optimize_weights(name):
    order, extra = "split_major", None
    g, target, sims = _faithful(name, order, extra)
    w, th, losses, last = so.faithful_optimize_weights(sims, g["labelled"], STREAMS, g["ballast"], 0.000003)
    assert {k: float(v) for k, v in w.items()} == g["opt_weights"]
    assert float(th) == g["opt_threshold"]
    assert [float(last[c]) for c in g["clip_order"]] == g["scores_after_optimize"]
    assert [float(v) for v in so.faithful_scores(sims, w).values()] == g["scores_opt"]
    if "border" in g:
        b = g["border"]
        w, th, _, _ = so.faithful_optimize_weights(sims, b["labelled"], STREAMS, b["ballast"], 0.000003)
        assert {k: float(v) for k, v in w.items()} == b["opt_weights"] and float(th) == b["opt_threshold"]


@pytest.mark.parametrize("name", ["synth_small", "ragged", "real_subset"])
def test_dense_restatement_matches_reference(name):
    """The vectorised oracle: dots differ from np.dot-on-lists only in summation order (<= 1e-15 here);
    everything downstream of the averaged similarities is bit-exact."""
    g, x, ids, present = _load(name)
    pos = {int(c): i for i, c in enumerate(ids)}
    rows = [pos[c] for c in g["clip_order"]]                             # DB rows in first-seen order
    x = x[rows]
    present = present[rows] if present is not None else None
    t = golden_target_array(g)
    sims, avg, n_e = so.dense_similarities(x, t, present)
    g_avg = np.array([[v if v is not None else np.nan for v in r] for r in g["sim_avg"]])
    assert (n_e == np.array(g["sim_n"])).all()
    ok = ~np.isnan(g_avg)
    assert np.abs(avg[ok] - g_avg[ok]).max() <= 2e-15
    if name != "ragged":
        # bit-exact downstream of the reference's own averaged similarities
        assert (so.dense_scores_libm(g_avg, [1.0, 1.5]) == np.array(g["scores_default"])).all()
        assert np.abs(so.dense_scores(g_avg, [1.0, 1.5]) - np.array(g["scores_default"])).max() <= 2.3e-16
        sc = np.array(g["scores_default"])
        m_rows, n_rows, amax = so.dense_select_partition(sc, 0.8, 0.35)
        assert [g["clip_order"][i] for i in m_rows] == [c for c, s in zip(g["clip_order"], sc) if s >= 0.8]
        lower = 0.8 - 0.35 * (1 - 0.8)
        near = [c for c, s in zip(g["clip_order"], sc) if lower <= s < 0.8]
        assert [g["clip_order"][i] for i in n_rows] == near
        if near:
            assert g["clip_order"][amax] == max(near, key=lambda c: sc[g["clip_order"].index(c)])
        # loss grid from grid scores (the vectorised form used beside the GPU grid kernel)
        lab = g["labelled"]
        ms = {}
        for m in lab:
            ms[m["video_clip"]] = m["user_match"] if m["user_match"] is not None else m["is_match"]
        lrows = [g["clip_order"].index(c) for c in ms]
        grid = np.stack([so.dense_scores_libm(g_avg[lrows], [1.0, w]) for w in so.WEIGHT_GRID])
        losses = so.dense_loss_grid(grid, np.array([ms[c] for c in ms]), g["ballast"])
        sims_dict = {c: {st: [g_avg[i, si], 3] for si, st in enumerate(STREAMS)} for i, c in enumerate(g["clip_order"])}
        _, _, ref_losses, _ = so.faithful_optimize_weights(sims_dict, lab, STREAMS, g["ballast"], 0.000003)
        assert (losses == ref_losses).all()


def test_cfg1_10k_dense_oracle():
    """BASELINE config[0]: 10k x 1024 (S=2, E=3); inputs regenerated from the seed."""
    meta = golden_json("cfg1_10k.json")
    z = np.load(__import__("os").path.join(__import__("_helpers").GOLDEN, "cfg1_10k.npz"))
    x = so.cfg1_features(n=10000, e=3, seed=0)
    assert hashlib.sha256(np.ascontiguousarray(x).tobytes()).hexdigest() == meta["inputs_sha256"], \
        "numpy RNG stream differs from the one the golden vectors were generated with"
    t = np.stack([[so.scale_feature(x[7, s, e].astype(np.float64)) for e in range(3)] for s in range(2)])
    assert hashlib.sha256(t.tobytes()).hexdigest() == meta["target_sha256"]
    _, avg, n_e = so.dense_similarities(x, t)
    assert np.abs(avg - z["sim_avg"]).max() <= 2e-15
    sc = so.dense_scores(avg, [1.0, 1.5])
    assert np.abs(sc - z["scores_default"]).max() <= 4e-15
    # the reference squares through libm pow (numpy scalar **): reproduce it bit for bit, and bound
    # the correctly-rounded variant (the one the HIP kernel matches bit for bit) at one ulp
    assert (so.dense_scores_libm(z["sim_avg"], [1.0, 1.5]) == z["scores_default"]).all()
    assert np.abs(so.dense_scores(z["sim_avg"], [1.0, 1.5]) - z["scores_default"]).max() <= 2.3e-16
    assert (np.argsort(-sc, kind="stable")[:50] == np.argsort(-z["scores_default"], kind="stable")[:50]).all()
    wopt = meta["opt_weights"]["warped_optical_flow"]
    assert (so.dense_scores_libm(z["sim_avg"], [1.0, wopt]) == z["scores_opt"]).all()
    assert (so.dense_scores_libm(z["sim_avg"], [1.0, so.WEIGHT_GRID[-1]]) == z["scores_after_optimize"]).all()
    assert np.abs(so.dense_scores(z["sim_avg"], [1.0, wopt]) - z["scores_opt"]).max() <= 2.3e-16


def test_synth_generator_is_deterministic_and_bounded():
    a = so.synth_features(3, 10, 4, 2, 5, 1024, (4.0, 1.0))
    b = so.synth_features(3, 12, 2, 2, 5, 1024, (4.0, 1.0))
    assert (a[2:] == b).all()
    assert a.dtype == np.float32 and a.min() >= 0 and a[:, 0].max() < 4.0 and a[:, 1].max() < 1.0
    assert 1.9 < a[:, 0].mean() < 2.1
