"""GPU: the HIP similarity / scoring / selection path (through the C ABI) against the oracle and the
golden vectors recorded from the reference.

Tolerances (stated once):
  * dot products / averaged similarities: |delta| <= 1e-12 (fp64 accumulate; only the summation order
    differs from numpy's ddot) -- observed ~1e-16.
  * scores GIVEN identical averaged similarities: bit-exact vs oracle.dense_scores; <= 2.3e-16 vs the
    reference (its ``**2`` is libm pow, see oracle.dense_scores docstring).
  * partitions, arg-max, top-k rows, ranks: identical.
"""
import numpy as np
import pytest

import sim_oracle as so
from _helpers import STREAMS, golden_json, golden_npy, golden_target_array

pytestmark = pytest.mark.gpu
SIM_TOL = 1e-12


@pytest.fixture(scope="module")
def vqa(gpu):
    import video_query_algorithms_amd as m
    return m


def _case(name):
    g = golden_json(name + ".json")
    x = golden_npy(name + "_x.npy")
    ids = np.asarray(g.get("clip_ids") or g["clip_order"])
    present = np.array(g["present"], dtype=np.uint8) if "present" in g else None
    pos = {int(c): i for i, c in enumerate(ids)}
    rows = [pos[c] for c in g["clip_order"]]
    return g, x[rows], np.asarray(g["clip_order"]), (present[rows] if present is not None else None)


@pytest.mark.parametrize("name", ["synth_small", "ragged", "real_subset"])
def test_scan_matches_reference_golden(vqa, name):
    g, x, ids, present = _case(name)
    db = vqa.FeatureDB.from_arrays(x, clip_ids=ids, present=present)       # fp32 or fp64 as stored
    t = golden_target_array(g)
    db.set_query(t)
    db.scan(weights=[1.0, 1.5], keep_sims=True)
    avg, n_e, sims = db.similarities(sims=True)
    o_sims, o_avg, o_ne = so.dense_similarities(x, t, present)
    assert (n_e == o_ne).all() and (n_e == np.array(g["sim_n"])).all()
    assert np.abs(sims - o_sims).max() <= SIM_TOL
    g_avg = np.array([[v if v is not None else np.nan for v in r] for r in g["sim_avg"]])
    ok = ~np.isnan(g_avg)
    assert np.abs(avg[ok] - g_avg[ok]).max() <= SIM_TOL
    assert np.isnan(avg[~ok]).all()
    # ensemble mean is bit-exact given the device's own per-split dots (sequential sum / count)
    pres = np.ones_like(n_e[..., None].repeat(x.shape[2], -1), dtype=bool) if present is None else present.astype(bool)
    acc = np.zeros_like(avg)
    for e in range(x.shape[2]):
        acc = np.where(pres[:, :, e], acc + sims[:, :, e], acc)
    with np.errstate(invalid="ignore", divide="ignore"):
        assert np.array_equal(acc / n_e, avg, equal_nan=True)
    if name != "ragged":
        sc = db.scores()
        assert (sc == so.dense_scores(avg, [1.0, 1.5])).all()             # bit-exact given the same avg
        assert np.abs(sc - np.array(g["scores_default"])).max() <= SIM_TOL
        assert (np.argsort(-sc, kind="stable") == np.argsort(-np.array(g["scores_default"]), kind="stable")).all()


def test_fp32_and_fp64_storage_agree_on_fp32_exact_inputs(vqa):
    g, x, ids, _ = _case("synth_small")
    assert x.dtype == np.float32
    t = golden_target_array(g)
    out = []
    for dt in (np.float32, np.float64):
        db = vqa.FeatureDB.from_arrays(x.astype(dt), clip_ids=ids)
        db.set_query(t)
        db.scan(weights=[1.0, 1.5])
        out.append((db.similarities()[0], db.scores()))
    assert (out[0][0] == out[1][0]).all() and (out[0][1] == out[1][1]).all()


def test_query_from_resident_row(vqa):
    g, x, ids, _ = _case("synth_small")
    db = vqa.FeatureDB.from_arrays(x, clip_ids=ids)
    row = list(ids).index(g["ref_clip_id"])
    t = db.set_query_from_row(row)
    want = golden_target_array(g)
    assert np.abs(t - want).max() <= 1e-18 + 4e-16 * np.abs(want).max()
    db.scan(weights=[1.0, 1.5])
    avg, _ = db.similarities()
    assert np.abs(avg[row] - 1.0).max() <= 1e-14                         # self-similarity is 1
    assert abs(db.scores()[row] - 1.0) <= 1e-14


def test_rescore_and_grid_are_bit_exact_given_avg(vqa):
    g, x, ids, _ = _case("real_subset")
    db = vqa.FeatureDB.from_arrays(x, clip_ids=ids)
    db.set_query(golden_target_array(g))
    db.scan()
    avg, _ = db.similarities()
    for w in ([1.0, 1.5], [1.0, 0.5], [1.0, 2.45], [0.3, 7.0]):
        db.rescore(w)
        assert (db.scores() == so.dense_scores(avg, w)).all()
    rows = [3, 0, 17, 5, 5, 23]
    wg = np.stack([np.ones(40), so.WEIGHT_GRID], axis=1)
    got = db.scores_grid(wg, rows)
    want = np.stack([so.dense_scores(avg[rows], w) for w in wg])
    assert (got == want).all()


def test_cfg1_10k_against_reference(vqa):
    """BASELINE config[0]: 10k x 1024, S=2, E=3, fp32 features; reference outputs from tests/golden."""
    import os
    from _helpers import GOLDEN
    z = np.load(os.path.join(GOLDEN, "cfg1_10k.npz"))
    meta = golden_json("cfg1_10k.json")
    x = so.cfg1_features(n=10000, e=3, seed=0)
    db = vqa.FeatureDB.from_arrays(x)
    t = np.stack([[so.scale_feature(x[7, s, e].astype(np.float64)) for e in range(3)] for s in range(2)])
    db.set_query(t)
    db.scan(weights=[1.0, 1.5])
    avg, n_e = db.similarities()
    assert (n_e == 3).all()
    assert np.abs(avg - z["sim_avg"]).max() <= SIM_TOL
    sc = db.scores()
    assert np.abs(sc - z["scores_default"]).max() <= SIM_TOL
    assert (np.argsort(-sc, kind="stable") == np.argsort(-z["scores_default"], kind="stable")).all()   # all 10k ranks
    m, r, amax = db.select(0.8, 0.8 - 0.35 * 0.2)
    assert [int(db.clip_ids[i]) for i in m] == [c for c, _ in meta["select_default"]]
    rows, vals = db.topk(100)
    o_rows, o_vals = so.dense_topk(z["scores_default"], 100)
    assert (rows == o_rows).all()
    db.rescore([1.0, meta["opt_weights"]["warped_optical_flow"]])
    assert np.abs(db.scores() - z["scores_opt"]).max() <= SIM_TOL


def _check_select(vqa, scores, th, lower):
    n = scores.shape[0]
    db = vqa.FeatureDB(n, 2, 1, 4)
    # plant the scores directly: avg such that score(w=[1,0]) = avg[:,0]
    avg = np.stack([scores, np.zeros(n)], axis=1)
    db.write_avg(avg, np.ones((n, 2), dtype=np.int32))
    db.rescore([1.0, 0.0])
    sc = db.scores()
    ok = ~np.isnan(scores)
    assert np.array_equal(sc[ok], 1.0 - np.sqrt(((1.0 - scores[ok]) ** 2) / 1.0))
    m, r, amax = db.select(th, lower)
    om, onr, oamax = so.dense_select_partition(sc, th, (th - lower) / (1 - th))
    # oracle takes near_miss; recompute its partition with the explicit lower bound instead
    om = np.flatnonzero(sc >= th)
    onr = np.flatnonzero((lower <= sc) & (sc < th))
    oamax = int(onr[np.argmax(sc[onr])]) if onr.size else -1
    assert np.array_equal(m, om) and np.array_equal(r, onr) and amax == oamax
    return db, sc


@pytest.mark.parametrize("n", [1, 7, 2048, 2049, 16383, 16384, 16385, 100003])
def test_select_partition_is_stable_and_complete(vqa, n):
    rng = np.random.default_rng(n)
    scores = rng.random(n)
    if n > 10:
        scores[rng.integers(0, n, 5)] = 0.8                  # exact ties on the threshold
        scores[rng.integers(0, n, 3)] = np.nan               # NaN belongs to neither class
        dup = rng.integers(0, n, 4)
        scores[dup] = 0.7999                                  # tied near-maximum: first one must win
    _check_select(vqa, scores, 0.8, 0.73)
    _check_select(vqa, scores, 2.0, 1.5)                     # empty classes
    _check_select(vqa, scores, -1.0, -2.0)                   # everything matches


@pytest.mark.parametrize("n,k", [(1, 1), (50, 50), (5000, 20), (16384, 1024), (9001, 1000), (16385, 20), (5000, 1025), (100003, 1000), (100003, 100003)])
def test_topk_sorted_stable(vqa, n, k):
    """n <= 16 384 and k <= 1 024: the one-workgroup kernel (keys in LDS); everything else: the radix-select launch sequence."""
    rng = np.random.default_rng(k)
    scores = np.round(rng.random(n), 3)                       # many exact ties
    if n > 10:
        scores[rng.integers(0, n, 3)] = np.nan
        scores[rng.integers(0, n, 3)] = -0.5
    db, sc = _check_select(vqa, scores, 0.8, 0.73)
    rows, vals = db.topk(k)
    valid = np.flatnonzero(~np.isnan(sc))
    order = valid[np.argsort(-sc[valid], kind="stable")][:k]
    assert np.array_equal(rows, order)
    assert np.array_equal(vals, sc[order])
    assert (np.diff(vals) <= 0).all()


def test_generic_shapes_fall_back_to_the_generic_kernel(vqa):
    rng = np.random.default_rng(5)
    for (n, s, e, d) in [(33, 3, 2, 64), (17, 2, 7, 1024), (9, 1, 1, 260), (40, 2, 3, 2048)]:
        x = np.abs(rng.standard_normal((n, s, e, d))).astype(np.float32)
        t = rng.standard_normal((s, e, d))
        present = rng.random((n, s, e)) > 0.2
        present[:, :, 0] = True
        db = vqa.FeatureDB.from_arrays(x, present=present)
        db.set_query(t)
        w = list(np.linspace(1.0, 2.0, s))
        db.scan(weights=w, keep_sims=True)
        avg, n_e, sims = db.similarities(sims=True)
        o_sims, o_avg, o_ne = so.dense_similarities(x, t, present)
        assert (n_e == o_ne).all()
        assert np.abs(sims - o_sims).max() <= 1e-11 and np.abs(avg - o_avg).max() <= 1e-11
        assert (db.scores() == so.dense_scores(avg, w)).all()


def test_synthetic_db_matches_host_generator_and_properties_at_scale(vqa):
    """cfg-4 shape on one GPU at reduced N (properties are size-independent): the device-generated DB equals
    the host regeneration on sampled slices; scan parity on those slices; linearity; permutation of weights."""
    n, s, e, d = 200_000, 2, 5, 1024
    scales = (4.0, 1.0)
    db = vqa.FeatureDB.synthetic(n, s, e, d, seed=17, scales=scales, row0=1000)
    t = db.set_query_from_row(12345)
    db.scan(weights=[1.0, 1.5])
    avg, n_e = db.similarities()
    sc = db.scores()
    assert (n_e == e).all()
    for row0 in (0, 4096, 123_456, n - 4096):
        x = so.synth_features(17, 1000 + row0, 4096, s, e, d, scales)
        _, o_avg, _ = so.dense_similarities(x, t)
        assert np.abs(avg[row0:row0 + 4096] - o_avg).max() <= SIM_TOL
    assert (sc == so.dense_scores(avg, [1.0, 1.5])).all()
    assert abs(sc[12345] - 1.0) <= 1e-14 and sc.argmax() == 12345
    # linearity of the scan in the query: sim(2t) = 2 sim(t) exactly (power-of-two scaling)
    db.set_query(2.0 * t)
    db.scan()
    assert (db.similarities()[0] == 2.0 * avg).all()
    # selection at scale agrees with numpy
    db.set_query(t)
    db.scan(weights=[1.0, 1.5])
    th = float(np.quantile(sc, 0.999))
    lower = float(np.quantile(sc, 0.99))
    m, r, amax = db.select(th, lower)
    assert np.array_equal(m, np.flatnonzero(sc >= th))
    assert np.array_equal(r, np.flatnonzero((lower <= sc) & (sc < th)))
    rows, vals = db.topk(20)
    o_rows, _ = so.dense_topk(sc, 20)
    assert np.array_equal(rows, o_rows)


def test_errors_are_reported_not_swallowed(vqa):
    db = vqa.FeatureDB(8, 2, 3, 1024)
    with pytest.raises(vqa.VqError) as ei:
        db.scan()
    assert ei.value.code == -4 and "query" in str(ei.value)
    with pytest.raises(vqa.VqError):
        vqa.FeatureDB(0, 2, 3, 1024)
    with pytest.raises(vqa.VqError):
        vqa.FeatureDB(8, 2, 3, 1022)
    with pytest.raises(vqa.VqError):
        db.scores()


def test_library_memory_is_visible_to_rccl_collectives(vqa):
    """The N > 1 path hands library-owned device memory to torch.distributed (backend nccl = RCCL) through a
    zero-copy __cuda_array_interface__ view.  One GPU here, so a 1-rank RCCL group: the collective must read the
    library's score buffer and reproduce it."""
    import os
    import socket
    import torch
    import torch.distributed as dist
    sys_path_bench = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    import sys
    sys.path.insert(0, sys_path_bench)
    from bench import dev_tensor
    db = vqa.FeatureDB.synthetic(4096, 2, 3, 1024, seed=3, scales=(4.0, 1.0))
    db.set_query_from_row(5, want=False)
    db.scan(weights=[1.0, 1.5])
    want = db.scores()
    dev = torch.device("cuda", 0)
    view = dev_tensor(db.scores_devptr(), (4096,), "<f8", dev)
    assert view.data_ptr() == db.scores_devptr()
    assert (view.cpu().numpy() == want).all()
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        out = torch.empty(4096, dtype=torch.float64, device=dev)
        dist.all_gather_into_tensor(out, view)
        torch.cuda.synchronize()
        assert (out.cpu().numpy() == want).all()
    finally:
        dist.destroy_process_group()


def test_db_from_binary_store_and_row_shards(vqa, tmp_path):
    """FeatureDB.from_store: chunked upload from the memory-mapped store, whole and as two row shards; scores equal the
    in-memory database bit for bit."""
    from video_query_algorithms_amd import feature_store as fs
    x = so.cfg1_features(n=300, e=3, seed=4)
    ids = np.arange(300) * 2 + 11
    present = np.ones((300, 2, 3), dtype=np.uint8)
    present[17, 0, 1] = 0
    present[250, 1, :2] = 0
    path = fs.save_store(str(tmp_path / "store"), x, ids, ("rgb", "warped_optical_flow"), (1, 2, 3), present=present)
    t = np.stack([[so.scale_feature(x[7, s, e].astype(np.float64)) for e in range(3)] for s in range(2)])
    ref = vqa.FeatureDB.from_arrays(x, clip_ids=ids, present=present)
    ref.set_query(t)
    ref.scan([1.0, 1.5])
    want = ref.scores()
    whole = vqa.FeatureDB.from_store(path, chunk_rows=64)
    whole.set_query(t)
    whole.scan([1.0, 1.5])
    assert (whole.scores() == want).all() and (whole.clip_ids == ids).all() and whole.stream_names == ["rgb", "warped_optical_flow"]
    got = []
    for row0, rows in ((0, 170), (170, 130)):
        part = vqa.FeatureDB.from_store(path, row0=row0, rows=rows, chunk_rows=50)
        part.set_query(t)
        part.scan([1.0, 1.5])
        got.append(part.scores())
        assert part.row_of(int(ids[row0])) == 0
        part.close()
    assert (np.concatenate(got) == want).all()
    with pytest.raises(ValueError):
        vqa.FeatureDB.from_store(path, row0=200, rows=200)
    ref.close()
    whole.close()


@pytest.mark.parametrize("dtype,q,n", [(np.float32, 1, 700), (np.float32, 3, 700), (np.float32, 8, 700), (np.float64, 5, 700),
                                       (np.float32, 16, 1003), (np.float64, 16, 33)])
def test_batched_scan_equals_single_scans_to_rounding(vqa, dtype, q, n):
    """vq_db_scan_batch: Q <= 16 queries in one pass over the database on the fp64 matrix cores (tiles of 16 clips x 16
    queries; ragged last tile, unused query slots, clips without some splits).  Per (query, clip) the dots, ensemble
    mean and score of the single-query scan up to the accumulation order: <= 1e-12 against it and against the oracle, the
    clip the query was made from scores 1, and every query's ranking equals the single scan's wherever scores differ by
    more than that tolerance."""
    x = so.cfg1_features(n=n, e=3, seed=6).astype(dtype)
    present = np.ones((n, 2, 3), dtype=np.uint8)
    present[5, 0, 2] = 0
    present[n // 6, 1, :2] = 0
    db = vqa.FeatureDB.from_arrays(x, present=present)
    rows = [(7 + 41 * i) % n for i in range(q)]
    targets = np.stack([np.stack([[so.scale_feature(x[r, s, e].astype(np.float64)) for e in range(3)] for s in range(2)]) for r in rows])
    weights = np.stack([[1.0, 1.5 + 0.1 * i] for i in range(q)])
    got = db.scan_batch(targets, weights)
    assert got.shape == (q, n) and np.isfinite(got).all()
    for i in range(q):
        db.set_query(targets[i])
        db.scan(weights[i])
        single = db.scores()
        assert np.abs(got[i] - single).max() <= 1e-12
        _, o_avg, _ = so.dense_similarities(x.astype(np.float64), targets[i], present.astype(bool))
        assert np.abs(got[i] - so.dense_scores(o_avg, weights[i])).max() <= 1e-12
        assert abs(got[i, rows[i]] - 1.0) <= 1e-12 or not present[rows[i]].all()
        order_b, order_s = np.argsort(-got[i], kind="stable"), np.argsort(-single, kind="stable")
        moved = order_b != order_s
        assert np.abs(single[order_b[moved]] - single[order_s[moved]]).max(initial=0.0) <= 2e-12     # only exact-tie neighbours may swap
    assert (db.scan_batch(targets, weights) == got).all()                                   # deterministic
    with pytest.raises(vqa.VqError):
        db.scan_batch(np.zeros((17, 2, 3, 1024)), np.ones((17, 2)))
    db.close()


@pytest.mark.parametrize("n,s,e,d,q,masked", [(140_000, 2, 1, 256, 16, False), (70_000, 2, 3, 256, 5, False), (5_000, 1, 2, 512, 16, True),
                                              (33, 2, 5, 1024, 16, False), (4_100, 3, 2, 768, 9, True)])
def test_fused_batched_scan_against_the_oracle(vqa, n, s, e, d, q, masked):
    """The single-launch batched scan (per-tile sums kept in registers across the slices, several rounds of tiles per
    workgroup, the chunk ring refilled across slice changes) against the oracle's dense restatement of ticket.py:151-180, query
    by query -- with and without a presence mask, 1 to 3 streams, ragged last tiles, a database large enough for a second round
    (140 000 clips > 256 workgroups x 16 waves x 2 tiles x 16 clips), every D the kernel takes.  Matrix-core accumulation order
    differs from numpy's: <= 1e-12 on the scores (observed ~1e-16); repeatable bit for bit."""
    db = vqa.FeatureDB.synthetic(n, s, e, d, seed=23, scales=(4.0, 1.0, 2.0)[:s])
    x = so.synth_features(23, 0, n, s, e, d, (4.0, 1.0, 2.0)[:s])
    present = None
    if masked:
        present = np.ones((n, s, e), dtype=np.uint8)
        present[::7, 0, 0] = 0
        present[3::11, s - 1, e - 1] = 0
        present[n - 1, :, 0] = 0
        present[..., 0] |= (present.sum(axis=2) == 0).astype(np.uint8)      # every (clip, stream) keeps at least one split
        db.set_present(present)
    rng = np.random.default_rng(n)
    targets = rng.standard_normal((q, s, e, d)) / d
    weights = 0.5 + rng.random((q, s))
    got = db.scan_batch(targets, weights)
    assert got.shape == (q, n) and np.isfinite(got).all()
    for k in range(q):
        _, avg, _ = so.dense_similarities(x, targets[k], present)
        assert np.abs(got[k] - so.dense_scores(avg, weights[k])).max() <= 1e-12, k
    assert (db.scan_batch(targets, weights) == got).all()
    db.close()


@pytest.mark.parametrize("n,s,e,q,masked", [(70_003, 2, 3, 16, False), (5_001, 1, 2, 7, True), (33, 2, 5, 16, False), (140_007, 2, 1, 16, True),
                                            (16, 1, 1, 1, False), (4_100, 2, 4, 9, True)])
def test_tiled_layout_in_place(vqa, n, s, e, q, masked):
    """vq_db_set_layout(TILED) turns the handle's block into [tile of 16 clips][slice][k / 4][clip][4] IN PLACE (no second copy).  On it:
    the 16-query pass has the bits of the pass on the rows (same operands per MFMA in the same order); the one-query scan
    (scan_tiled_kernel: a wave per tile) agrees with the row-major scan and with the oracle to <= 1e-12 and is bit-reproducible;
    rows read back, the query made from a row, uploads into the middle and the ragged end, and target bootstrapping see the same
    data; converting back restores the block bit for bit -- ragged last tiles, presence masks, 1-2 streams, 1-5 splits."""
    d = 1024
    rng = np.random.default_rng(n + e)
    targets = rng.standard_normal((q, s, e, d)) / d
    weights = 0.5 + rng.random((q, s))
    present = None
    if masked:
        present = np.ones((n, s, e), dtype=np.uint8)
        present[::5, 0, 0] = 0
        present[n - 1, :, e - 1] = 0
        present[..., 0] |= (present.sum(axis=2) == 0).astype(np.uint8)
    scales = (4.0, 1.0)[:s]
    plain = vqa.FeatureDB.synthetic(n, s, e, d, seed=29, scales=scales)
    db = vqa.FeatureDB.synthetic(n, s, e, d, seed=29, scales=scales)
    for h in (plain, db):
        h.set_present(present)
    assert db.layout == "rows"
    db.set_layout("tiled")
    assert db.layout == "tiled" and plain.layout == "rows"
    pick = np.array(sorted({0, n - 1, n // 2, min(17, n - 1), (n // 16) * 16 - 1 if n >= 16 else 0}))
    assert (db.read_rows(pick) == plain.read_rows(pick)).all()

    def both(fn):
        return fn(plain), fn(db)
    want, got = both(lambda h: h.scan_batch(targets, weights))
    assert (got == want).all() and (db.scan_batch(targets, weights) == got).all()
    t_rows, t_tiled = both(lambda h: h.set_query_from_row(int(pick[-1])))
    assert (t_rows == t_tiled).all()
    for h in (plain, db):
        h.scan(weights=[1.0, 1.5][:s], keep_sims=True)
    (avg_r, ne_r, sims_r), (avg_t, ne_t, sims_t) = both(lambda h: h.similarities(sims=True))
    x = so.synth_features(29, 0, n, s, e, d, scales) if n <= 6000 else None
    assert (ne_r == ne_t).all() and np.abs(avg_r - avg_t).max() <= 1e-12 and np.abs(sims_r - sims_t).max() <= 1e-12
    if x is not None:
        o_sims, o_avg, o_ne = so.dense_similarities(x, t_rows, present)
        assert (ne_t == o_ne).all() and np.abs(avg_t - o_avg).max() <= 1e-12 and np.abs(sims_t - o_sims).max() <= 1e-12
    sc_t = db.scores()
    assert (sc_t == so.dense_scores(avg_t, [1.0, 1.5][:s])).all()           # scores bit-exact given the averages, as on the rows
    db.scan(weights=[1.0, 1.5][:s], keep_sims=True)
    assert (db.scores() == sc_t).all() and (db.similarities()[0] == avg_t).all()
    if n >= 40 and not masked:
        v, iv = [int(pick[1]), 3, 20], [5, 30]
        b_r, b_t = both(lambda h: h.bootstrap_target(v, iv, mu=0.3, set_query=False))
        assert (b_r == b_t).all()
    # new rows in the middle and at the ragged end, through vq_db_upload: dealt into their tiles
    rows = (rng.random((40, s, e, d)) * 3).astype(np.float32)
    for h in (plain, db):
        h.upload(max(0, n // 2 - 10), rows[:min(25, n)])
        h.upload(max(0, n - 15), rows[25:25 + min(15, n)])
    want2, got2 = both(lambda h: h.scan_batch(targets, weights))
    assert not (want2 == want).all() and (got2 == want2).all()
    with pytest.raises(vqa.VqError):
        db.feats_devptr()                                              # a tiled block is not [N][S][E][D]
    db.set_layout("rows")
    assert (db.read_rows(np.arange(min(n, 64))) == plain.read_rows(np.arange(min(n, 64)))).all()
    assert (db.read_rows(np.arange(max(0, n - 40), n)) == plain.read_rows(np.arange(max(0, n - 40), n))).all()
    for h in (plain, db):
        h.set_query(t_rows)
        h.scan(weights=[1.0, 1.5][:s])
    assert (plain.scores() == db.scores()).all()                       # back on the rows: the row-major scan's own bits
    plain.close()
    db.close()


def test_tiled_layout_is_refused_where_it_cannot_hold(vqa):
    """fp64 databases, shapes without a tiled scan kernel, and blocks somebody else may write (adopted memory, a handed-out
    pointer) stay row-major: VQ_E_UNSUPPORTED / VQ_E_STATE, and the handle keeps working."""
    import torch
    rng = np.random.default_rng(8)
    for kw in ({"dim": 256}, {"dtype": np.float64}, {"n_streams": 3}):
        args = dict(n=200, n_streams=2, n_splits=2, dim=1024, dtype=np.float32)
        args.update(kw)
        db = vqa.FeatureDB.synthetic(args["n"], args["n_streams"], args["n_splits"], args["dim"], seed=3, scales=(2.0, 1.0, 1.0)[:args["n_streams"]],
                                     dtype=args["dtype"])
        with pytest.raises(vqa.VqError) as ei:
            db.set_layout("tiled")
        assert ei.value.code == -5 and db.layout == "rows"
        db.close()
    n, s, e, d = 2_000, 2, 2, 1024
    db = vqa.FeatureDB.synthetic(n, s, e, d, seed=31, scales=(2.0, 1.0))
    targets = rng.standard_normal((4, s, e, d)) / d
    weights = 0.5 + rng.random((4, s))
    first = db.scan_batch(targets, weights)
    ptr = db.feats_devptr()                                            # from here on others may write the rows
    with pytest.raises(vqa.VqError) as ei:
        db.set_layout("tiled")
    assert ei.value.code == -4
    new = (rng.random((n, s, e, d)) * 2).astype(np.float32)
    src = torch.from_numpy(new).cuda()
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    assert hip.hipMemcpy(C.c_void_p(ptr), C.c_void_p(src.data_ptr()), C.c_size_t(new.nbytes), 3) == 0     # hipMemcpyDeviceToDevice
    torch.cuda.synchronize()
    fresh = vqa.FeatureDB(n, s, e, d)
    fresh.upload(0, new)
    want = fresh.scan_batch(targets, weights)
    assert not (want == first).all() and (db.scan_batch(targets, weights) == want).all()     # writes through the pointer are seen
    fresh.close()
    db.close()


def test_cfg4_full_size_scan_one_million_clips(vqa):
    """BASELINE configs[3] at FULL size on one GPU: 1 query x 1 000 000 clips x 2 streams x 5 splits x 1024 fp32
    (40.96 GB resident, generated on the device), the real ``scan_kernel<float,2,5,4>`` launch of the bench.
    Parity on 4 x 4096-row oracle slices (regenerated on the host from the counter-based seed); size-independent
    properties over all rows: scores bit-exact given the averages, the reference clip scores 1 and is the arg-max,
    top-k / partition equal numpy on the full array, and a second scan reproduces every bit."""
    n, s, e, d = 1_000_000, 2, 5, 1024
    scales = (4.0, 1.0)
    db = vqa.FeatureDB.synthetic(n, s, e, d, seed=23, scales=scales)
    ref_row = 765_432
    t = db.set_query_from_row(ref_row)
    db.scan(weights=[1.0, 1.5])
    avg, n_e = db.similarities()
    sc = db.scores()
    assert (n_e == e).all() and np.isfinite(sc).all()
    for row0 in (0, 333_333, ref_row - 100, n - 4096):
        x = so.synth_features(23, row0, 4096, s, e, d, scales)
        _, o_avg, _ = so.dense_similarities(x, t)
        assert np.abs(avg[row0:row0 + 4096] - o_avg).max() <= SIM_TOL
    assert (sc == so.dense_scores(avg, [1.0, 1.5])).all()
    assert abs(sc[ref_row] - 1.0) <= 1e-14 and sc.argmax() == ref_row
    rows, vals = db.topk(20)
    o_rows, o_vals = so.dense_topk(sc, 20)
    assert np.array_equal(rows, o_rows) and (vals == o_vals).all()
    th, lower = float(np.quantile(sc, 0.9999)), float(np.quantile(sc, 0.999))
    m, r, amax = db.select(th, lower)
    assert np.array_equal(m, np.flatnonzero(sc >= th)) and np.array_equal(r, np.flatnonzero((lower <= sc) & (sc < th)))
    assert amax == int(r[np.argmax(sc[r])])
    db.scan(weights=[1.0, 1.5])
    assert (db.scores() == sc).all() and (db.similarities()[0] == avg).all()
    # the 16-query pass at this size (62 500 tiles: about eight rounds per workgroup) on the rows, then the SAME block tiled in place
    # (no second copy: 41 GB stay 41 GB): every query within 1e-12 of a single scan on the four oracle slices, the two layouts
    # bit-identical, and the one-query scan on the tiled block within 1e-12 of the row-major one everywhere
    q_rows = [ref_row] + [12_345 + 61_111 * i for i in range(15)]
    tb = np.stack([db.set_query_from_row(r) for r in q_rows])
    wb = np.stack([[1.0, 1.5 + 0.05 * i] for i in range(16)])
    on_rows = db.scan_batch(tb, wb)
    assert np.abs(on_rows[0] - sc).max() <= 1e-12 and np.isfinite(on_rows).all()
    for row0 in (0, 333_333, ref_row - 100, n - 4096):
        x = so.synth_features(23, row0, 4096, s, e, d, scales)
        for qi in (0, 7, 15):
            _, o_avg, _ = so.dense_similarities(x, tb[qi])
            assert np.abs(on_rows[qi, row0:row0 + 4096] - so.dense_scores(o_avg, wb[qi])).max() <= 1e-12
    assert all(abs(on_rows[i, r] - 1.0) <= 1e-12 and on_rows[i].argmax() == r for i, r in enumerate(q_rows))
    db.set_layout("tiled")
    assert (db.scan_batch(tb, wb) == on_rows).all()
    db.set_query(t)
    db.scan(weights=[1.0, 1.5])
    avg_t = db.similarities()[0]
    assert np.abs(avg_t - avg).max() <= SIM_TOL and (db.scores() == so.dense_scores(avg_t, [1.0, 1.5])).all()
    sc_t = db.scores()
    assert np.array_equal(db.topk(20)[0], so.dense_topk(sc_t, 20)[0]) and sc_t.argmax() == ref_row
    db.close()


def test_a_topk_between_select_and_fetch_is_detected(vqa):
    """Handles are shared between broker threads (broker.py:91-92).  The two-call form of the selection
    (vq_db_select, then vq_db_select_fetch) must not hand out another call's rows: a top-k in between invalidates
    the lists (VQ_E_STATE); the one-call form FeatureDB.select uses is atomic under the handle's lock."""
    import ctypes as C
    from video_query_algorithms_amd._lib import call
    rng = np.random.default_rng(4)
    # a small database's top-k (one workgroup, own scratch) leaves the selection lists alone: the fetch then returns them, correct
    small = vqa.FeatureDB(5000, 2, 1, 1024)
    sc = rng.random(5000)
    small.write_avg(np.stack([sc, sc], axis=1), np.ones((5000, 2), dtype=np.int32))
    small.rescore([1.0, 1.0])
    got = small.scores()
    nm, nn, am = C.c_int64(), C.c_int64(), C.c_int64()
    call("vq_db_select", small._h, 0.7, 0.6, C.byref(nm), C.byref(nn), C.byref(am))
    small.topk(10)
    m = np.empty(nm.value, dtype=np.int64)
    r = np.empty(nn.value, dtype=np.int64)
    call("vq_db_select_fetch", small._h, m.ctypes.data_as(C.c_void_p), m.size, r.ctypes.data_as(C.c_void_p), r.size)
    assert np.array_equal(m, np.flatnonzero(got >= 0.7)) and np.array_equal(r, np.flatnonzero((0.6 <= got) & (got < 0.7)))
    small.close()
    db = vqa.FeatureDB(20000, 2, 1, 1024)
    sc = rng.random(20000)
    avg = np.stack([sc, sc], axis=1)
    db.write_avg(avg, np.ones((20000, 2), dtype=np.int32))
    db.rescore([1.0, 1.0])
    got = db.scores()
    nm, nn, am = C.c_int64(), C.c_int64(), C.c_int64()
    call("vq_db_select", db._h, 0.7, 0.6, C.byref(nm), C.byref(nn), C.byref(am))
    db.topk(10)
    m = np.empty(nm.value, dtype=np.int64)
    r = np.empty(nn.value, dtype=np.int64)
    with pytest.raises(vqa.VqError) as ei:
        call("vq_db_select_fetch", db._h, m.ctypes.data_as(C.c_void_p), m.size, r.ctypes.data_as(C.c_void_p), r.size)
    assert ei.value.code == -4
    m2, r2, _ = db.select(0.7, 0.6)
    assert np.array_equal(m2, np.flatnonzero(got >= 0.7)) and np.array_equal(r2, np.flatnonzero((0.6 <= got) & (got < 0.7)))
    # hammer one handle from three threads: selections and top-k interleave, every selection stays self-consistent
    import threading
    errors = []

    def selector(th):
        try:
            for _ in range(30):
                a, b, _ = db.select(th, th - 0.1)
                if not (np.array_equal(a, np.flatnonzero(got >= th)) and np.array_equal(b, np.flatnonzero((th - 0.1 <= got) & (got < th)))):
                    errors.append("select(%g) returned foreign rows" % th)
        except Exception as exc:       # noqa: BLE001
            errors.append(repr(exc))

    def ranker():
        try:
            for _ in range(30):
                rows, _ = db.topk(50)
                if not np.array_equal(rows, so.dense_topk(got, 50)[0]):
                    errors.append("topk")
        except Exception as exc:       # noqa: BLE001
            errors.append(repr(exc))
    threads = [threading.Thread(target=selector, args=(0.9,)), threading.Thread(target=selector, args=(0.5,)), threading.Thread(target=ranker)]
    for th_ in threads:
        th_.start()
    for th_ in threads:
        th_.join()
    assert not errors, errors[:3]
    db.close()


def test_comm_group_of_the_c_abi_single_rank(vqa):
    """The Comm exports (vq_comm_*, vq_allgather_*, vq_broadcast_query) with a world of one rank on this one GPU: the
    RCCL communicator comes up from a unique id, the score slice of a sharded scan lands zero-padded in its slot, a
    feature block is gathered byte for byte.  (world > 1 needs one GPU per rank: the driver's multi-GPU run.)"""
    import ctypes as C
    import torch
    from video_query_algorithms_amd._lib import call
    dev = torch.device("cuda", 0)
    uid = (C.c_char * 128)()
    call("vq_comm_unique_id", uid)
    comm = C.c_void_p()
    call("vq_comm_init", 0, 1, uid, 0, C.byref(comm))
    rank, world, device = C.c_int32(-1), C.c_int32(-1), C.c_int32(-1)
    call("vq_comm_info", comm, C.byref(rank), C.byref(world), C.byref(device))
    assert (rank.value, world.value, device.value) == (0, 1, 0)
    db = vqa.FeatureDB.synthetic(3000, 2, 3, 1024, seed=3, scales=(4.0, 1.0))
    db.set_query_from_row(5, want=False)
    db.scan(weights=[1.0, 1.5])
    want = db.scores()
    out = torch.full((3072,), -1.0, dtype=torch.float64, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    call("vq_allgather_scores", comm, db._h, 3072, C.c_void_p(out.data_ptr()), C.c_void_p(stream))
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    assert (got[:3000] == want).all() and (got[3000:] == 0).all()
    block = torch.randint(0, 256, (1 << 20,), dtype=torch.uint8, device=dev)
    allb = torch.zeros_like(block)
    call("vq_allgather_features", comm, C.c_void_p(block.data_ptr()), block.numel(), C.c_void_p(allb.data_ptr()), C.c_void_p(stream))
    q = torch.arange(2 * 3 * 1024, dtype=torch.float64, device=dev)
    call("vq_broadcast_query", comm, C.c_void_p(q.data_ptr()), q.numel() * 8, 0, C.c_void_p(stream))
    torch.cuda.synchronize()
    assert torch.equal(allb, block) and torch.equal(q.cpu(), torch.arange(2 * 3 * 1024, dtype=torch.float64))
    with pytest.raises(vqa.VqError):
        call("vq_allgather_scores", comm, db._h, 100, C.c_void_p(out.data_ptr()), C.c_void_p(stream))   # slice too small
    # The stream contract (vq_amd.h): the scan is only ENQUEUED on the database handle's stream; the gather may run on any other
    # stream and still sees the scores of the scan just enqueued (an event orders the copy behind it) -- here the scan of a new
    # query on the handle's own non-blocking stream, the gather at once on another one, no synchronisation in between.
    s_scan, s_gather = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    db.set_stream(s_scan.cuda_stream)
    for row in (77, 1234, 5):
        db.set_query_from_row(row, want=False)
        db.scan(weights=[1.0, 1.5])
        out.fill_(-1.0)
        torch.cuda.synchronize()
        db.set_query_from_row(row, want=False)
        db.scan(weights=[0.5, 2.5])                      # enqueued ...
        call("vq_allgather_scores", comm, db._h, 3072, C.c_void_p(out.data_ptr()), C.c_void_p(s_gather.cuda_stream))   # ... and gathered at once
        s_gather.synchronize()
        got = out.cpu().numpy()
        assert (got[:3000] == db.scores()).all() and abs(got[row] - 1.0) <= 1e-12
    # a handle that holds no scores (a new query, no scan yet) is refused, not gathered stale
    db.set_query_from_row(9, want=False)
    with pytest.raises(vqa.VqError) as ei:
        call("vq_allgather_scores", comm, db._h, 3072, C.c_void_p(out.data_ptr()), C.c_void_p(stream))
    assert ei.value.code == -4                           # VQ_E_STATE
    db.set_stream(None)
    call("vq_comm_destroy", comm)
    db.close()


def test_the_raw_ctypes_stub_of_the_one_call_round(vqa):
    """INTEGRATION.md 1, "the whole round as one call": the stub a maintainer with their own wrapper binds -- raw ctypes on the C ABI, a
    page-locked block laid out as vq_db_round_layout says -- against the package's own objects."""
    import ctypes as C
    n = 3000
    rng = np.random.default_rng(8)
    feats = np.abs(rng.standard_normal((n, 2, 3, 1024))).astype(np.float32)
    t = feats[5].astype(np.float64)
    t = t / (t * t).sum(axis=-1, keepdims=True)
    ref = vqa.FeatureDB.from_arrays(feats)
    ref.set_query(t)
    ref.scan(weights=[1.0, 1.5])
    want_avg, _ = ref.similarities()
    want_scores = ref.scores()
    want_match, want_near, want_am = ref.select(0.8, 0.73)
    ref.close()
    lib = C.CDLL(vqa._lib.LIB_PATH)
    db = C.c_void_p()
    assert lib.vq_db_create(C.c_int64(n), 2, 3, 1024, 0, 0, C.byref(db)) == 0
    assert lib.vq_db_upload(db, C.c_int64(0), C.c_int64(n), feats.ctypes.data_as(C.c_void_p)) == 0
    off = (C.c_int64 * 10)()
    assert lib.vq_db_round_layout(db, off) == 0
    blk = C.c_void_p()
    assert lib.vq_host_alloc(C.byref(blk), C.c_int64(off[8])) == 0
    raw = np.ctypeslib.as_array(C.cast(blk, C.POINTER(C.c_uint8)), (off[8],))
    raw[off[0]:off[0] + t.nbytes].view(np.float64)[:] = t.ravel()
    raw[off[1]:off[1] + 16].view(np.float64)[:] = (1.0, 1.5)
    assert lib.vq_db_query_round(db, blk, C.c_int64(off[8]), 1 | 2 | 4, C.c_double(0.8), C.c_double(0.73)) == 0
    avg = raw[off[2]:off[2] + n * 2 * 8].view(np.float64).reshape(n, 2)
    scores = raw[off[4]:off[4] + n * 8].view(np.float64)
    n_match, n_near, near_argmax = raw[off[5]:off[5] + 24].view(np.int64)
    match_rows = raw[off[6]:off[6] + 8 * min(n_match, off[9])].view(np.int64)
    near_rows = raw[off[7]:off[7] + 8 * min(n_near, off[9])].view(np.int64)
    assert (avg == want_avg).all() and (scores == want_scores).all()
    assert (n_match, n_near, near_argmax) == (len(want_match), len(want_near), want_am)
    assert (match_rows == want_match).all() and (near_rows == want_near).all()
    # a block that is too small, or a selection without scores: refused with a message, nothing launched
    assert lib.vq_db_query_round(db, blk, C.c_int64(off[8] - 1), 7, C.c_double(0.8), C.c_double(0.73)) == -1
    assert lib.vq_db_query_round(db, blk, C.c_int64(off[8]), 1 | 4, C.c_double(0.8), C.c_double(0.73)) == -1
    lib.vq_last_error.restype = C.c_char_p
    assert b"selection needs the scores" in lib.vq_last_error()
    assert lib.vq_host_free(blk) == 0 and lib.vq_db_destroy(db) == 0
