"""Test program (GPU box): a ShardedFeatureDB over RCCL with ONE rank and a database large enough that a scan takes milliseconds
(300 000 rows x 2 streams x 3 splits x 1024 fp32 = 7.4 GB).  The database's stream is a non-blocking side stream and vq_db_scan
only queues its launch: a host read that is not ordered behind that stream returns what the PREVIOUS query left in the result
arrays (with the 64-clip fixtures of the other sharded tests a scan is over before the host gets to read, whatever the order).
Every result of a second query must equal a plain one-GPU FeatureDB's bit for bit.  Prints ``ok``."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 300_000
    import torch
    import torch.distributed as dist
    import video_query_algorithms_amd as vqa
    from video_query_algorithms_amd.sharded_db import ShardedFeatureDB
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29577")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        sdb = ShardedFeatureDB.synthetic(n, 2, 3, 1024, seed=5)
        one = vqa.FeatureDB.synthetic(n, 2, 3, 1024, seed=5)
        for layout in ("rows", "tiled"):
            if layout == "tiled":
                sdb.set_layout("tiled")
                one.set_layout("tiled")
            want = {}
            for row, w in ((11, [1.0, 1.5]), (n // 2 + 3, [1.0, 0.7])):       # the second query is the one a stale read would miss
                t = one.set_query_from_row(row)
                one.scan(weights=w)
                ts = sdb.set_query_from_row(row)
                assert (ts == t).all()
                sdb.scan(weights=w)                                          # queued on the side stream; the reads below follow at once
                avg_s, ne_s = sdb.similarities()
                sc_s = sdb.scores()
                grid = np.stack([np.ones(8), np.linspace(0.5, 2.25, 8)], axis=1)
                rows = np.array([0, 5, row, n - 1, n // 3])
                g_s = sdb.scores_grid(grid, rows)
                m_s = sdb.min_score([3, row, n - 2])
                sel_s = sdb.select(0.8, 0.7)
                top_s = sdb.topk(20)
                avg_o, ne_o = one.similarities()
                assert (avg_s == avg_o).all() and (ne_s == ne_o).all(), (layout, row, "similarities")
                assert (sc_s == one.scores()).all(), (layout, row, "scores")
                assert (g_s == one.scores_grid(grid, rows)).all(), (layout, row, "grid")
                assert m_s == one.min_score([3, row, n - 2])
                sel_o = one.select(0.8, 0.7)
                assert all((np.asarray(x) == np.asarray(y)).all() for x, y in zip(sel_s[:2], sel_o[:2])) and sel_s[2] == sel_o[2]
                top_o = one.topk(20)
                assert (top_s[0] == top_o[0]).all() and (top_s[1] == top_o[1]).all()
                want[row] = sc_s
            assert not (want[11] == want[n // 2 + 3]).all()                   # the two queries really differ
        sdb.close()
        one.close()
    finally:
        dist.destroy_process_group()
    print("ok", flush=True)


if __name__ == "__main__":
    main()
