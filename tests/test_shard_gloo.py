"""CPU, 2 processes over gloo: the partition / all-gather / merge logic used for N > 1 GPUs.  The ranks'
"kernels" here are the numpy oracle (no GPU in this container); what is under test is that sharded results
reassemble to exactly the unsharded ones, in order."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_range_is_a_partition():
    from video_query_algorithms_amd.shard import shard_range
    for n in (1, 7, 8, 1000, 1_000_000, 10_001):
        for world in (1, 2, 3, 4, 8):
            spans = [shard_range(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and sum(c for _, c in spans) == n
            for (a0, ac), (b0, _) in zip(spans, spans[1:]):
                assert a0 + ac == b0
            assert max(c for _, c in spans) - min(c for _, c in spans) <= 1


def test_merge_topk_matches_global_stable_sort():
    from video_query_algorithms_amd.shard import merge_topk, shard_range
    rng = np.random.default_rng(0)
    scores = np.round(rng.random(1003), 2)                    # many ties across shards
    world, k = 4, 25
    rows, vals, r0s = [], [], []
    for r in range(world):
        r0, cnt = shard_range(scores.size, world, r)
        local = scores[r0:r0 + cnt]
        o = np.argsort(-local, kind="stable")[:k]
        rows.append(o)
        vals.append(local[o])
        r0s.append(r0)
    grow, gval = merge_topk(rows, vals, r0s, k)
    want = np.argsort(-scores, kind="stable")[:k]
    assert (grow == want).all() and (gval == scores[want]).all()


def _worker(rank, world, port, n_total, tmp):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sim_oracle as so
    from video_query_algorithms_amd.shard import all_gather_rows, shard_range
    s, e, d = 2, 3, 64
    r0, cnt = shard_range(n_total, world, rank)
    x = so.synth_features(5, r0, cnt, s, e, d, (4.0, 1.0))           # this rank's DB rows (same generator as the GPU)
    t = torch.zeros((s, e, d), dtype=torch.float64)
    if rank == 0:
        full0 = so.synth_features(5, 0, 8, s, e, d, (4.0, 1.0))
        t.copy_(torch.from_numpy(np.stack([[so.scale_feature(full0[3, si, ei].astype(np.float64)) for ei in range(e)]
                                           for si in range(s)])))
    dist.broadcast(t, 0)                                              # query + weights are the only broadcast
    _, avg, _ = so.dense_similarities(x, t.numpy())
    local_scores = torch.from_numpy(so.dense_scores(avg, [1.0, 1.5]))
    scores = all_gather_rows(local_scores, n_total)                   # score slices -> every rank
    feats = all_gather_rows(torch.from_numpy(x.reshape(cnt, -1)), n_total)   # feature blocks (A -> B hand-off)
    np.save(os.path.join(tmp, "scores_%d.npy" % rank), scores.numpy())
    np.save(os.path.join(tmp, "feats_%d.npy" % rank), feats.numpy())
    dist.destroy_process_group()


@pytest.mark.parametrize("n_total", [64, 37])
def test_two_rank_gather_equals_unsharded(tmp_path, n_total):
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import sim_oracle as so
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    mp.spawn(_worker, args=(2, port, n_total, str(tmp_path)), nprocs=2, join=True)
    s, e, d = 2, 3, 64
    x = so.synth_features(5, 0, n_total, s, e, d, (4.0, 1.0))
    t = np.stack([[so.scale_feature(x[3, si, ei].astype(np.float64)) for ei in range(e)] for si in range(s)])
    _, avg, _ = so.dense_similarities(x, t)
    want = so.dense_scores(avg, [1.0, 1.5])
    for rank in range(2):
        got = np.load(tmp_path / ("scores_%d.npy" % rank))
        assert got.shape == (n_total,) and (got == want).all()          # each clip's score is independent of sharding
        f = np.load(tmp_path / ("feats_%d.npy" % rank))
        assert (f == x.reshape(n_total, -1)).all()                      # global clip order preserved
