"""GPU: target bootstrapping (csrc/vq_boot.hip through the C ABI) against
  * the CPU oracle's explicit-inverse restatement of target_clip.py:192-197 / :245-260, and
  * the targets and downstream scores the REFERENCE ITSELF produced (tests/golden/bootstrap*, oracle/gen_golden_bootstrap.py).

Tolerance (fp64, stated once): the device solves two small systems on the Gram matrix of the validated rows, the
reference inverts 1024 x 1024 matrices; both are backward stable, they differ by rounding amplified by the
conditioning of X M^-1 X^T (1e3 .. 1e6 on these fixtures).  |dw| <= 1e-8 max|w| on every target, scores within 1e-9,
and the ranking of the clips by score identical to the reference's (up to swaps among clips whose reference scores
tie within that tolerance: the validated matches all score 1 +- 1e-12 by construction)."""
import json
import os
import random

import numpy as np
import pytest

import bootstrap_oracle as bo
from _helpers import DEFAULT_WEIGHTS, GOLDEN, SEED, STREAMS, golden_json, golden_npy, records_from_dense

pytestmark = pytest.mark.gpu
SPLITS = (1, 2, 3)
W_TOL = 1e-8
S_TOL = 1e-9


@pytest.fixture(scope="module")
def vqa(gpu):
    import video_query_algorithms_amd as m
    return m


@pytest.mark.parametrize("m,n,mu,dtype", [(6, 0, 0.0, np.float64), (6, 4, 0.0, np.float64), (6, 4, 0.3, np.float64),
                                          (3, 7, 1.5, np.float64), (1, 1, 0.3, np.float64), (1, 0, 0.0, np.float64),
                                          (10, 9, 0.05, np.float64), (8, 5, 0.3, np.float32), (20, 30, 0.3, np.float32)])
def test_closed_forms_match_the_explicit_inverses(vqa, m, n, mu, dtype):
    from video_query_algorithms_amd.bootstrap import bootstrap_targets
    if dtype == np.float64:
        x = golden_npy("real_subset_x.npy")
        X, Y = x[:m, 0, 1], x[12:12 + n, 0, 1]
        X2, Y2 = x[2:2 + m, 1, 2], x[10:10 + n, 1, 0]
    else:
        x = golden_npy("synth_small_x.npy")
        X, Y = x[:m, 0, 1], x[30:30 + n, 0, 1]
        X2, Y2 = x[5:5 + m, 1, 2], x[28:28 + n, 1, 0]
    got = bootstrap_targets([(X, Y if n else None), (X2, Y2 if n else None)], mu)
    for w, (A, B) in zip(got, [(X, Y), (X2, Y2)]):
        A64, B64 = A.astype(np.float64), B.astype(np.float64)
        want = bo.bootstrap_valid_invalid(A64, B64, mu) if n else bo.bootstrap_valid(A64)
        assert np.abs(w - want).max() <= W_TOL * np.abs(want).max()
        assert np.abs(A64 @ w - 1).max() <= 1e-7            # every validated match scores 1 against the new target


def test_ragged_problem_sizes_in_one_launch(vqa):
    from video_query_algorithms_amd.bootstrap import bootstrap_targets
    x = golden_npy("real_subset_x.npy")
    probs = [(x[:3, 0, 0], None), (x[:5, 1, 1], x[10:12, 1, 1]), (x[4:5, 0, 2], x[9:20, 0, 2])]
    got = bootstrap_targets(probs, 0.3)
    for w, (A, B) in zip(got, probs):
        want = bo.bootstrap_valid(A) if B is None else bo.bootstrap_valid_invalid(A, B, 0.3)
        assert np.abs(w - want).max() <= W_TOL * np.abs(want).max()


def test_singular_system_raises_like_linalg_inv(vqa):
    from video_query_algorithms_amd.bootstrap import bootstrap_targets
    x = golden_npy("real_subset_x.npy")
    dup = np.stack([x[0, 0, 0], x[0, 0, 0] * 0.0])            # a zero row: X X^T is exactly singular
    with pytest.raises(vqa.VqError):
        bootstrap_targets([(dup, None)], 0.0)


def _case(vqa, name):
    meta = golden_json("bootstrap.json")[name]
    targets = np.load(os.path.join(GOLDEN, "bootstrap_targets.npz"), allow_pickle=False)
    if name.startswith("synth"):
        x = golden_npy("synth_small_x.npy")
        ids = np.arange(1, x.shape[0] + 1) * 3 + 100
        ref = int(ids[7])
    else:
        x = golden_npy("real_subset_x.npy")
        ids = np.array(golden_json("real_subset.json")["clip_ids"])
        ref = 10
    prev = None
    if meta["had_previous"]:
        p = targets["previous_target"]
        prev = {st: {sp: p[si, ei].tolist() for ei, sp in enumerate(SPLITS)} for si, st in enumerate(STREAMS)}
    tk = vqa.Ticket({"query_id": 1, "video_id": 1, "ref_clip": 0, "ref_clip_id": ref, "search_set": 1,
                     "dynamic_target_adjustment": True,
                     "latest_query_result": {"id": 5, "round": 1, "bootstrapped_target": prev},
                     "match_list": meta["matches"], "match_page_size": 4},
                    records=records_from_dense(x, ids, list(SPLITS)))
    hp = vqa.Hyperparameter(DEFAULT_WEIGHTS, 0.8, 0.0, 0.35, meta["mu"], STREAMS, "global_pool", meta["f_bootstrap"], meta["f_memory"],
                            meta["bootstrap_type"], meta["nbags"])
    return meta, targets[name], tk, hp


@pytest.mark.parametrize("name", ["simple_valid", "simple_valid_half", "simple_both_mu0", "simple_both_mu03", "simple_both_mu03_half",
                                  "partial_both", "partial_noprev", "bagging_valid", "bagging_both_mu0", "bagging_both_mu03",
                                  "no_valid_matches", "synth_bagging_mu03", "synth_simple_mu03"])
def test_target_clip_round_matches_reference(vqa, name):
    """The whole TargetClip.get_target_features of a round >= 2 (same seed => same random draws), then the scan with
    the bootstrapped target: targets, scores and ranking against the reference's."""
    meta, want, tk, hp = _case(vqa, name)
    tk.target = vqa.TargetClip(tk, hp)
    random.seed(a=SEED)
    tk.target.get_target_features()
    got = np.array([[tk.target.target_features[st][sp] for sp in SPLITS] for st in STREAMS], dtype=np.float64)
    assert np.abs(got - want).max() <= W_TOL * np.abs(want).max()
    assert isinstance(tk.target.target_features["rgb"][1], list)          # stays JSON-serialisable (ticket.py:296)
    json.dumps(tk.target.target_features)
    tk.compute_similarities(hp)
    tk.compute_scores(DEFAULT_WEIGHTS)
    assert list(tk.scores.keys()) == meta["clip_order"]
    s_got = np.array([tk.scores[c] for c in meta["clip_order"]])
    s_want = np.array(meta["scores_default"])
    assert np.abs(s_got - s_want).max() <= S_TOL
    # same ranking.  The validated matches all score 1 up to rounding (in the reference too), so clips whose reference
    # scores differ by less than the tolerance may swap; everything else must be in the reference's order.
    order = np.argsort(-s_got, kind="stable")
    assert (np.diff(s_want[order]) <= 2 * S_TOL).all()
    distinct = np.abs(np.diff(np.sort(s_want))).min() > 2 * S_TOL
    if distinct:
        assert order.tolist() == np.argsort(-s_want, kind="stable").tolist()


def test_resident_db_bootstrap(vqa):
    """vq_db_bootstrap_target: validated clips addressed as rows of a resident FeatureDB; the result becomes the
    query of the next scan without a host round trip."""
    x = golden_npy("real_subset_x.npy")
    db = vqa.FeatureDB.from_arrays(x)
    valid, invalid = [1, 2, 5, 8], [12, 13, 20]
    t = db.bootstrap_target(valid, invalid, mu=0.3)
    for si in range(2):
        for ei in range(3):
            want = bo.bootstrap_valid_invalid(x[valid, si, ei], x[invalid, si, ei], 0.3)
            assert np.abs(t[si, ei] - want).max() <= W_TOL * np.abs(want).max()
    db.scan([1.0, 1.5])
    avg, _ = db.similarities()
    assert np.abs(avg[valid] - 1.0).max() <= 1e-7              # validated matches now sit at similarity 1
    t0 = db.bootstrap_target(valid, mu=0.3, set_query=False)
    for si in range(2):
        for ei in range(3):
            want = bo.bootstrap_valid(x[valid, si, ei])
            assert np.abs(t0[si, ei] - want).max() <= W_TOL * np.abs(want).max()
    with pytest.raises(vqa.VqError):
        db.bootstrap_target([999], mu=0.0)
    db.close()
