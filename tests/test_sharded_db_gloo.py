"""CPU, 2-3 processes over gloo: the row-sharded database behind the B seam (sharded_db.ShardedFeatureDB).

Every rank's "kernels" are the numpy oracle (tests/_standin_db.py: no GPU in this container); under test is everything
ShardedFeatureDB adds -- announcing operations to serving ranks, broadcasting the query, gathering avg / n_e / score slices into
global first-seen order, the rank-major concatenation of the per-rank threshold partitions, the top-k merge, the error
agreement -- driven through the SAME Ticket / Hyperparameter seam methods the broker calls, against what the reference itself
produced (tests/golden, oracle/gen_golden.py).  The GPU twin is tests/test_sharded_db_gpu.py."""
import os
import random
import socket
import sys

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOL = 1e-12


def _port():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _setup(rank, world, port):
    for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("COMPUTE_EPS", "0.000003")
    dist.init_process_group("gloo", rank=rank, world_size=world)


def _golden_block(name):
    """(golden, features [N,S,E,D] in first-seen order, present or None, clip ids in that order)."""
    from _helpers import golden_json, golden_npy
    g = golden_json(name + ".json")
    x = golden_npy(name + "_x.npy")
    ids = list(g.get("clip_ids") or g["clip_order"])
    order = [ids.index(c) for c in g["clip_order"]]
    present = np.array(g["present"], dtype=bool)[order] if "present" in g else None
    return g, np.ascontiguousarray(x[order]), present, np.asarray(g["clip_order"], dtype=np.int64)


def _sharded(x, present, ids, served, fail_first_scan_on=None):
    from _helpers import STREAMS
    from _standin_db import OracleFeatureDB
    from video_query_algorithms_amd.shard import shard_range
    from video_query_algorithms_amd.sharded_db import ShardedFeatureDB
    rank, world = dist.get_rank(), dist.get_world_size()
    r0, cnt = shard_range(x.shape[0], world, rank)
    local = OracleFeatureDB(x[r0:r0 + cnt], None if present is None else present[r0:r0 + cnt], list(STREAMS), [[1, 2, 3]] * 2,
                            fail_on_scan=(fail_first_scan_on == rank))
    return ShardedFeatureDB(local, x.shape[0], r0, ids, served=served)


def _close(a, b, tol=TOL):
    return abs(float(a) - float(b)) <= tol


def _same_matches(got, want):
    got = [[int(k), float(v)] for k, v in got.items()]
    assert [k for k, _ in got] == [k for k, _ in want], (got, want)
    assert all(_close(a[1], b[1]) for a, b in zip(got, want))


def _broker_round(name, sdb):
    """Rank 0 of the served database: the assertions of tests/test_ticket_gpu.py::test_query_round_matches_reference, through
    classes patched by install()."""
    import video_query_algorithms_amd as vqa
    from _helpers import DEFAULT_WEIGHTS, SEED, STREAMS, records_from_dense
    g, x, present, ids = _golden_block(name)
    ref_row = list(ids).index(g["ref_clip_id"])
    ref_records = records_from_dense(x[ref_row:ref_row + 1], [g["ref_clip_id"]], [1, 2, 3],
                                     None if present is None else present[ref_row:ref_row + 1])

    class BrokerTicket:                                   # a look-alike of the reference's Ticket: install() grafts the hot path
        def __init__(self):
            self.ref_clip_id, self.search_set = g["ref_clip_id"], 1
            self.user_matches = g.get("user_matches", {})
            self.dynamic_target_adjustment, self.latest_query_result = False, None
            self.matches = []

        def _request(self, action, params):
            assert action == ["video-clips", "features"]          # with a resident database the search set is never downloaded
            return ref_records if params["id"] == g["ref_clip_id"] else []

    class BrokerHp(vqa.Hyperparameter):
        pass
    vqa.install(BrokerTicket, BrokerHp)
    assert BrokerTicket.compute_similarities is vqa.TicketScoring.compute_similarities
    tk = BrokerTicket()
    tk.feature_db = sdb
    hp = BrokerHp(DEFAULT_WEIGHTS, 0.8, 0.0, 0.35, 0.0, STREAMS, "global_pool", 1, 0.7, "bagging", 3)
    tk.target = vqa.TargetClip(tk, hp)
    tk.target.get_target_features()
    assert sorted(tk.target.splits) == g["target_splits"]
    announced = []
    real_announce = type(sdb)._announce

    def counting(self, op, ints=(), floats=()):
        announced.append(op)
        return real_announce(self, op, ints, floats)
    type(sdb)._announce = counting
    try:
        tk.compute_similarities(hp)
        tk.compute_scores(DEFAULT_WEIGHTS)
        random.seed(a=SEED)
        tk.select_clips_to_review(0.8, 20, 0.35)
    finally:
        type(sdb)._announce = real_announce
    from video_query_algorithms_amd.sharded_db import OP_RESTRICT, OP_ROUND
    assert [op for op in announced if op != OP_RESTRICT] == [OP_ROUND], announced      # the whole round: ONE announcement to the ranks
    _same_matches(tk.matches, g["select_default"])
    assert list(tk.similarities.keys()) == g["clip_order"]
    for c, avg_row, n_row in zip(g["clip_order"], g["sim_avg"], g["sim_n"]):
        entry = tk.similarities[c]
        for si, st in enumerate(STREAMS):
            if n_row[si]:
                assert _close(entry[st][0], avg_row[si]) and entry[st][1] == n_row[si]
            else:
                assert st not in entry
    tk.compute_scores(DEFAULT_WEIGHTS)
    assert list(tk.scores.keys()) == g["clip_order"]
    assert all(_close(tk.scores[c], s) for c, s in zip(g["clip_order"], g["scores_default"]))
    random.seed(a=SEED)
    tk.select_clips_to_review(0.8, 20, 0.35)
    _same_matches(tk.matches, g["select_default"])
    random.seed(a=SEED)
    tk.select_clips_to_review(0.8, 6, 0.5)
    _same_matches(tk.matches, g["select_max6"])
    low, clip = tk.lowest_scoring_user_match()
    assert _close(low, g["lowest_user_match"][0]) and clip == g["lowest_user_match"][1]
    near = max(0.8 - low, 0) / max(1 - 0.8, 0.000003)
    random.seed(a=SEED)
    tk.select_clips_to_review(0.8, float("inf"), near)
    _same_matches(tk.matches, g["select_finalize"])
    if "opt_weights" in g and present is None:            # the weight update: 40-weight grid over the labelled rows, sharded
        hp2 = BrokerHp(DEFAULT_WEIGHTS, 0.8, g["ballast"], 0.35, 0.0, STREAMS, "global_pool", 1, 0.7, "bagging", 3)
        tk.matches = g["labelled"]
        hp2.optimize_weights(tk)
        for k in g["opt_weights"]:
            assert _close(hp2.weights[k], g["opt_weights"][k], 1e-9)
        assert _close(hp2.threshold, g["opt_threshold"], 1e-9)
        tk.compute_scores(dict(g["opt_weights"]))
        assert all(_close(tk.scores[c], s) for c, s in zip(g["clip_order"], g["scores_opt"]))


def _shared_rounds(name, sdb):
    """Several tickets -- and threads -- on the ONE served database, and a badly shaped query before a normal
    round (it must be refused BEFORE the announcement, or the workers wait for a partner that never comes)."""
    import video_query_algorithms_amd as vqa
    from _helpers import DEFAULT_WEIGHTS, STREAMS, records_from_dense
    from _round_threads import check_shared_database
    g, x, _present, ids = _golden_block(name)
    with pytest.raises(ValueError):
        sdb.set_query(np.zeros((sdb.S, sdb.E, sdb.D + 1)))
    with pytest.raises(ValueError):
        sdb.write_avg(np.zeros((sdb.n + 1, sdb.S)))
    recs = records_from_dense(x, ids, [1, 2, 3])
    check_shared_database(vqa, sdb, recs, [int(c) for c in np.asarray(ids)[[0, 7, 19, 33]]], g["labelled"], STREAMS, DEFAULT_WEIGHTS,
                          threads=2, repeats=2)


def _served_worker(rank, world, port, name):
    _setup(rank, world, port)
    _g, x, present, ids = _golden_block(name)
    sdb = _sharded(x, present, ids, served=True)
    try:
        if rank == 0:
            _broker_round(name, sdb)
            if name == "synth_small":
                _shared_rounds(name, sdb)
            sdb.close()
        else:
            sdb.serve()                                   # returns when rank 0 closes the database
            assert sdb.local.closed is False
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("name,world", [("synth_small", 2), ("ragged", 2), ("real_subset", 3)])
def test_query_round_through_install_on_a_served_sharded_database(name, world):
    mp.spawn(_served_worker, args=(world, _port(), name), nprocs=world, join=True)


def _spmd_worker(rank, world, port, n_total, tmp):
    _setup(rank, world, port)
    import sim_oracle as so
    s, e, d = 2, 3, 64
    x = so.synth_features(5, 0, n_total, s, e, d, (4.0, 1.0))
    x[n_total // 2:, :, :, :] = x[: n_total - n_total // 2]           # duplicated clips: exact score ties ACROSS the shards
    ids = np.arange(100, 100 + n_total)
    sdb = _sharded(x, None, ids, served=False)
    t = sdb.set_query_from_row(n_total - 2)                           # the last rank holds the row
    want_t = np.stack([[so.scale_feature(x[n_total - 2, si, ei].astype(np.float64)) for ei in range(e)] for si in range(s)])
    assert (t == want_t).all()
    sdb.scan(weights=[1.0, 1.5], keep_sims=True)
    avg, ne, sims = sdb.similarities(sims=True)
    o_sims, o_avg, o_ne = so.dense_similarities(x, want_t)
    assert (avg == o_avg).all() and (ne == o_ne).all() and (sims == o_sims).all()     # every clip independent of the sharding: bit for bit
    assert all((a == b).all() for a, b in zip(sdb.similarities(), (o_avg, o_ne)))
    sc = sdb.scores()
    o_sc = so.dense_scores(o_avg, [1.0, 1.5])
    assert (sc == o_sc).all()
    th = float(np.sort(o_sc)[-n_total // 4])
    m, r, am = sdb.select(th, th - 0.2)
    om, orr, oam = so.dense_select_partition(o_sc, th, 0.2 / (1 - th))
    assert (m == om).all() and (r == np.flatnonzero((th - 0.2 <= o_sc) & (o_sc < th))).all()
    near = np.flatnonzero((th - 0.2 <= o_sc) & (o_sc < th))
    assert am == (int(near[np.argmax(o_sc[near])]) if near.size else -1)      # FIRST maximum in global order
    rows, vals = sdb.topk(9)
    orows, ovals = so.dense_topk(o_sc, 9)
    assert (rows == orows).all() and (vals == ovals).all()
    pick = np.array([n_total - 1, 0, n_total // 2, 3])
    assert sdb.min_score(pick) == min(1.0, o_sc[pick].min())
    assert (sdb.read_rows(pick) == x[pick]).all()
    wg = np.stack([np.ones(5), np.linspace(0.5, 2.5, 5)], 1)
    assert (sdb.scores_grid(wg, pick) == np.stack([so.dense_scores(o_avg[pick], w) for w in wg])).all()
    sdb.rescore([1.0, 0.7])
    assert (sdb.scores() == so.dense_scores(o_avg, [1.0, 0.7])).all()
    tb = np.stack([want_t, want_t * 0.5])
    wb = np.array([[1.0, 1.5], [1.0, 2.0]])
    got = sdb.scan_batch(tb, wb)
    assert got.shape == (2, n_total) and (got[0] == o_sc).all()
    # the round as ONE operation: the arrays of the separate calls above, bit for bit -- with the scan ...
    r1 = sdb.query_round(want_t, weights=[1.0, 1.5], select=(th, th - 0.2))
    assert (r1.avg == o_avg).all() and (r1.n_e == o_ne).all() and r1.n_e.dtype == np.int32 and (r1.scores == o_sc).all()
    assert (r1.match_rows == m).all() and (r1.near_rows == r).all() and r1.near_argmax == am
    # ... and as a re-weighting of the similarities the ranks hold (no scan, no similarities back)
    r2 = sdb.query_round(None, weights=[1.0, 0.7], select=(th, th - 0.2))
    sc2 = so.dense_scores(o_avg, [1.0, 0.7])
    assert r2.avg is None and (r2.scores == sc2).all() and (r2.match_rows == np.flatnonzero(sc2 >= th)).all()
    assert (r2.near_rows == np.flatnonzero((th - 0.2 <= sc2) & (sc2 < th))).all() and (sdb.scores() == sc2).all()
    with pytest.raises(ValueError):
        sdb.query_round(None, weights=None, select=(th, th - 0.2))      # refused before anything is announced
    sdb.set_query(want_t * 2.0 if rank == 0 else None)                # SPMD: the root's vectors win
    sdb.scan(weights=None)
    assert (sdb.similarities()[0] == so.dense_similarities(x, want_t * 2.0)[1]).all()
    assert sdb.row_of(100 + n_total - 1) == n_total - 1 and sdb.has_clip(100) and not sdb.has_clip(5)
    np.save(os.path.join(tmp, "ok_%d.npy" % rank), np.zeros(1))
    sdb.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_total,world", [(64, 2), (37, 3)])
def test_spmd_sharded_results_equal_unsharded(tmp_path, n_total, world):
    mp.spawn(_spmd_worker, args=(world, _port(), n_total, str(tmp_path)), nprocs=world, join=True)
    assert all(os.path.exists(tmp_path / ("ok_%d.npy" % r)) for r in range(world))


def _failing_worker(rank, world, port):
    _setup(rank, world, port)
    import sim_oracle as so
    from video_query_algorithms_amd.sharded_db import ShardError
    x = so.synth_features(5, 0, 32, 2, 3, 64, (4.0, 1.0))
    sdb = _sharded(x, None, np.arange(1, 33), served=True, fail_first_scan_on=1)
    if rank == 0:
        sdb.set_query_from_row(3)
        with pytest.raises(ShardError, match=r"rank\(s\) \[1\]"):
            sdb.scan(weights=[1.0, 1.5])                               # rank 1's kernel "fails": the broker gets an exception,
        sdb.scan(weights=[1.0, 1.5])                                   # not a hang, and the database keeps serving
        t = np.stack([[so.scale_feature(x[3, si, ei].astype(np.float64)) for ei in range(3)] for si in range(2)])
        assert (sdb.scores() == so.dense_scores(so.dense_similarities(x, t)[1], [1.0, 1.5])).all()
        sdb.close()
    else:
        scan = sdb.local.scan

        def once(*a, **k):                                             # fail the first scan only
            sdb.local.fail_on_scan, sdb.local.scan = False, scan
            raise RuntimeError("stand-in: this rank's scan fails")
        sdb.local.scan = once
        sdb.serve()
    dist.destroy_process_group()


def test_a_failing_rank_raises_on_the_broker_and_the_database_keeps_serving():
    mp.spawn(_failing_worker, args=(2, _port()), nprocs=2, join=True)


def test_fanout_counts_gpus_without_hip(monkeypatch):
    from video_query_algorithms_amd import fanout
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "2,5,7")
    assert fanout.visible_gpus() == 3
    monkeypatch.delenv("HIP_VISIBLE_DEVICES")
    for var in ("ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        monkeypatch.delenv(var, raising=False)
    assert fanout.visible_gpus() >= 0                                   # KFD topology (0 in a container without /sys/class/kfd)
    assert "torch" not in fanout.visible_gpus.__code__.co_names


def test_a_terminated_parent_takes_its_children_with_it(tmp_path):
    """fanout.run_children: the per-GPU children hold GPUs and may sit in a collective; a parent that gets SIGTERM (or Ctrl-C, or an
    exception) must end them, not orphan them."""
    import signal
    import subprocess
    import time
    child = tmp_path / "child.py"
    child.write_text("import os, sys, time\nopen(sys.argv[1] + '.' + os.environ['RANK'], 'w').write(str(os.getpid()))\ntime.sleep(600)\n")
    parent = tmp_path / "parent.py"
    parent.write_text("import sys\nsys.path.insert(0, %r)\nfrom video_query_algorithms_amd import fanout\n"
                      "sys.exit(fanout.run_children(%r, [%r], fanout.rank_envs(2)))\n" % (ROOT, str(child), str(tmp_path / "pid")))
    p = subprocess.Popen([sys.executable, str(parent)])
    deadline = time.time() + 120
    while time.time() < deadline and not all((tmp_path / ("pid.%d" % r)).exists() for r in range(2)):
        time.sleep(0.1)
    pids = [int((tmp_path / ("pid.%d" % r)).read_text()) for r in range(2)]
    p.send_signal(signal.SIGTERM)
    p.wait(timeout=60)
    time.sleep(0.5)
    for pid in pids:
        try:
            os.kill(pid, 0)
            alive = True
        except ProcessLookupError:
            alive = False
        assert not alive, "child %d outlived its parent" % pid


def test_open_fails_fast_when_a_worker_cannot_start(tmp_path):
    """ShardedFeatureDB.open watches its workers while it waits for them: a worker that dies at start (here: the store path does not
    exist) is a ShardError in the broker within seconds, not a rendezvous that runs into its 600 s timeout."""
    import subprocess
    import time
    code = ("import sys, time\nsys.path.insert(0, %r)\n"
            "from video_query_algorithms_amd.sharded_db import ShardedFeatureDB, ShardError\n"
            "t0 = time.time()\n"
            "try:\n    ShardedFeatureDB.open(%r, gpus=[0, 1], backend='gloo')\n"
            "except ShardError as e:\n    print('ShardError after %%.1f s: %%s' %% (time.time() - t0, e))\n" % (ROOT, str(tmp_path / "no_such_store")))
    t0 = time.time()
    r = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert "ShardError after" in r.stdout and "exited with code" in r.stdout, r.stdout[-2000:]
    assert time.time() - t0 < 120
