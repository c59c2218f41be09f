"""The row-sharded feature database behind the B seam: N GPUs of one node score ONE query.

The reference scores a query in one process (``Ticket.compute_similarities`` / ``compute_scores`` /
``select_clips_to_review``, src/models/ticket.py:120-180,311-356, called from compute_matches.py:58,77,89).
:class:`ShardedFeatureDB` has the surface of :class:`FeatureDB` that ``TicketScoring``, ``Hyperparameter`` and
``TargetClip`` use, so ``vqa.install(Ticket, ...)`` is unchanged -- attach one to ``ticket.feature_db`` and the same
broker code scores on every GPU of the node (half B):

* the database is row-sharded: rank g holds the contiguous rows ``shard_range(N, world, g)`` of the ``[N][S][E][D]``
  block as an ordinary one-GPU :class:`FeatureDB`; rank-major order = global first-seen order;
* the query (``S*E*D`` doubles = 80 KB at cfg 4) and the weights are the only things sent TO the ranks;
* every rank scans its own rows with the same kernels as the one-GPU path (``csrc/vq_sim.hip``), so a clip's
  similarity and score do not depend on the sharding -- bit for bit;
* what comes back is ``avg`` / ``n_e`` / score slices (N x 8 bytes per array -- never features), all-gathered into
  global order (``shard.all_gather_rows``: one fixed-size ``all_gather_into_tensor``, RCCL over xGMI);
* the threshold partition of ``select_clips_to_review`` runs on every rank's slice (the stable partition kernel) and
  the row lists are concatenated rank-major = order-preserving; ``random.sample`` then draws on ONE rank, in the
  broker's process, so a seeded broker picks the clips CPython would (ticket.py:333,341);
* top-k: per-rank top-k lists merged with the global stable tie-break (``shard.merge_topk``);
* target bootstrapping: the handful of user-validated rows are fetched from their owners (``vq_db_read_rows``) and the
  closed forms run on the root's GPU (``csrc/vq_boot.hip``, the same kernel the resident route uses).

Two ways to run it.  **SPMD** (``bench.py``, tests): every rank calls the same methods with the same arguments.
**Served** (the broker): the broker's process is rank 0 and calls the methods; the other ranks sit in
:meth:`serve` and are told what to do by a 3-word header + payload broadcast -- ``ShardedFeatureDB.open(store,
gpus=[...])`` starts them as fresh processes (``shard_worker.py``), one per GPU.  An exception on any rank is agreed on
before the next collective (served mode) and raised on the root, like a one-GPU ``VqError`` would be.

Nothing here computes on the host: the arithmetic is the local :class:`FeatureDB`'s.
"""
from __future__ import annotations

import functools
import os
import sys
import threading
from typing import Optional, Sequence

import numpy as np

from .shard import all_gather_rows, merge_topk, shard_range

OP_CLOSE, OP_RESTRICT, OP_SET_QUERY, OP_QUERY_FROM_ROW, OP_SCAN, OP_RESCORE, OP_SIMS, OP_SCORES, OP_GRID, OP_SELECT, OP_TOPK, \
    OP_MIN, OP_FETCH, OP_SCAN_BATCH, OP_LAYOUT, OP_WRITE_AVG, OP_ROUND = range(17)


def _atomic(method):
    """One operation of the sharded database = an announcement, this rank's step and its collectives: two threads of the broker's
    process must not interleave them (the workers would pair one operation's header with the other's payload).  The lock is
    re-entrant, so a caller that holds it across a whole round (TicketScoring) nests."""
    @functools.wraps(method)
    def locked(self, *args, **kw):
        with self.lock:
            return method(self, *args, **kw)
    return locked


class ShardError(RuntimeError):
    """A rank of the sharded database failed; the message names the rank(s) and carries the root's own error if any."""


class ShardedFeatureDB:
    """``local``: this rank's rows ``[row0, row0 + local.n)`` as a :class:`FeatureDB` (or an object with its local
    surface).  ``clip_ids``: the GLOBAL id list (every rank holds it: N x 8 bytes).  ``served``: rank ``root`` drives,
    the others must be inside :meth:`serve`."""

    def __init__(self, local, n_total: int, row0: int, clip_ids: Sequence[int], group=None, root: int = 0, served: bool = False, stream=None):
        if hasattr(local, "_h"):                          # a real FeatureDB (the CPU tests' stand-in has no library handle)
            from . import _lib
            _lib.require_torch_runtime("ShardedFeatureDB")
        import torch
        import torch.distributed as dist
        self._torch, self._dist = torch, dist
        self.local, self.group, self.root, self.served = local, group, int(root), bool(served)
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.n, self.row0 = int(n_total), int(row0)
        self.S, self.E, self.D, self.dtype = local.S, local.E, local.D, np.dtype(local.dtype)
        self.device = getattr(local, "device", 0)
        self.clip_ids = np.asarray(clip_ids, dtype=np.int64)
        if self.clip_ids.shape != (self.n,):
            raise ValueError("clip_ids must list all %d clips of the database" % self.n)
        if (self.row0, local.n) != shard_range(self.n, self.world, self.rank):
            raise ValueError("rank %d must hold rows %r of %d, got [%d,+%d)" % (self.rank, shard_range(self.n, self.world, self.rank),
                                                                              self.n, self.row0, local.n))
        self.stream_names = getattr(local, "stream_names", None)
        self.slot_splits = getattr(local, "slot_splits", None)
        self._row_of = None
        # as FeatureDB: a re-entrant lock for groups of calls, and whose similarities / scores the ranks hold right now
        self.lock = threading.RLock()
        self.sims_owner = None
        self.scores_owner = None
        backend = dist.get_backend(group)
        # collectives move tensors on the backend's device: RCCL = this rank's GPU (result arrays are viewed in place, never
        # staged through the host), gloo = host memory (CPU tests; rehearsals of N ranks on one card)
        self._cdev = torch.device("cuda", self.device) if backend == "nccl" else torch.device("cpu")
        self._stream = None
        if self._cdev.type == "cuda":
            self._stream = stream if stream is not None else torch.cuda.Stream(device=self._cdev)
            local.set_stream(self._stream.cuda_stream)           # scans and collectives are ordered on one stream
        flag = torch.tensor([0 if getattr(local, "present", None) is None else 1], dtype=torch.int64, device=self._cdev)
        self._coll(lambda: dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=group))
        # what TargetClip._resident_ok asks: None = every (clip, stream, split) is there, on every rank
        self.present = None if int(self._host(flag)[0]) == 0 else "sharded"
        self._closed = False

    # ------------------------------------------------------------------ construction
    @classmethod
    def from_store(cls, path: str, device: int = 0, group=None, root: int = 0, served: bool = False, chunk_rows: int = 16384):
        """Every rank uploads ITS row range of a binary feature store (feature_store.py) -- the files are memory-mapped, so a
        rank reads only its own N/world rows."""
        import torch.distributed as dist
        from .feature_db import FeatureDB
        from .feature_store import open_store
        _meta, feats, ids, _present = open_store(path)
        row0, rows = shard_range(feats.shape[0], dist.get_world_size(group), dist.get_rank(group))
        local = FeatureDB.from_store(path, device=device, row0=row0, rows=rows, chunk_rows=chunk_rows)
        return cls(local, feats.shape[0], row0, ids, group=group, root=root, served=served)

    @classmethod
    def synthetic(cls, n: int, n_streams: int, n_splits: int, dim: int = 1024, seed: int = 0, scales=(4.0, 1.0), dtype=np.float32,
                  device: int = 0, group=None, root: int = 0, stream=None):
        """Rows generated on each rank's device by the counter-based hash (BASELINE configs[3]: too big to ship)."""
        import torch.distributed as dist
        from .feature_db import FeatureDB
        row0, rows = shard_range(n, dist.get_world_size(group), dist.get_rank(group))
        local = FeatureDB.synthetic(rows, n_streams, n_splits, dim, seed=seed, scales=scales, row0=row0, dtype=dtype, device=device)
        return cls(local, n, row0, np.arange(1, n + 1, dtype=np.int64), group=group, root=root, stream=stream)

    @classmethod
    def open(cls, store_path: str, gpus: Sequence[int], backend: Optional[str] = None, timeout_s: float = 600.0):
        """The broker's entry point: start one fresh worker process per further GPU (``shard_worker.py``; call this BEFORE the
        broker's process makes its first GPU call), join them as rank 0 over 127.0.0.1 and return the served database on
        ``gpus[0]``.  ``close()`` ends the workers."""
        import datetime
        import torch
        import torch.distributed as dist
        from . import fanout
        gpus = [int(g) for g in gpus]
        if not gpus:
            raise ValueError("no GPU named")
        if dist.is_initialized():
            raise RuntimeError("ShardedFeatureDB.open starts its own process group; this process already has one "
                               "(build the database with from_store(group=...) instead)")
        backend = backend or os.environ.get("VQ_DIST_BACKEND") or "nccl"
        world = len(gpus)
        envs = fanout.rank_envs(world)
        port = int(envs[0]["MASTER_PORT"])
        # the rendezvous store is this process's own: the workers report through it ("ready" once their arguments and the store
        # files check out, "loaded" once their rows are on their GPU), and while it waits the broker watches that they are alive --
        # a worker that dies on a typo or on a full GPU is an exception here within a second, not a collective that times out
        store = dist.TCPStore("127.0.0.1", port, world, True, datetime.timedelta(seconds=timeout_s), wait_for_workers=False)
        procs = []
        if world > 1:
            worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "shard_worker.py")
            argv = [os.path.abspath(store_path), "--backend", backend, "--timeout", str(timeout_s)]
            for e, g in zip(envs[1:], gpus[1:]):
                e["VQ_FANOUT_DEVICE"] = str(g)
            procs = fanout.start_children(worker, argv, envs[1:])
        try:
            _await_workers(store, ["ready/%d" % r for r in range(1, world)], procs, timeout_s, "start")
            kw = {"rank": 0, "world_size": world, "store": store, "timeout": datetime.timedelta(seconds=timeout_s)}
            if backend == "nccl":
                torch.cuda.set_device(gpus[0])
                kw["device_id"] = torch.device("cuda", gpus[0])
            dist.init_process_group(backend, **kw)
            from .feature_db import FeatureDB
            from .feature_store import open_store
            _meta, feats, ids, _present = open_store(store_path)
            row0, rows = shard_range(feats.shape[0], world, 0)
            local = FeatureDB.from_store(store_path, device=gpus[0], row0=row0, rows=rows)
            _await_workers(store, ["loaded/%d" % r for r in range(1, world)], procs, timeout_s, "load their rows")
            db = cls(local, feats.shape[0], row0, ids, served=True)
        except BaseException:
            fanout.stop_children(procs)
            if dist.is_initialized():
                dist.destroy_process_group()
            raise
        db._workers, db._owns_group = procs, True
        return db

    # ------------------------------------------------------------------ plumbing
    def _coll(self, fn):
        if self._stream is None:
            return fn()
        with self._torch.cuda.stream(self._stream):
            return fn()

    def _host(self, t) -> np.ndarray:
        """A collective's (or the library's) device tensor as a host array.  The copy is issued ON the database's stream: scans only
        queue their launch there (vq_db_scan does not synchronise) and the stream is non-blocking, so a ``.cpu()`` / ``.item()`` on
        torch's null stream would not wait for the scan or the collective that produced the values -- it would read what the
        previous round left behind."""
        return self._coll(lambda: t.cpu()).numpy()

    @property
    def _driving(self) -> bool:
        return self.served and self.rank == self.root

    def _announce(self, op: int, ints=(), floats=()):
        """Served mode, root only: tell the ranks inside serve() what comes next.  One int64 header [op, #ints, #floats] and one
        int64 payload (the doubles travel as their bit patterns)."""
        if not self._driving:
            return
        gone = [(i + 1, p.returncode) for i, p in enumerate(getattr(self, "_workers", ())) if p.poll() is not None]
        if gone and op != OP_CLOSE:                      # a worker process that has died cannot answer: say so instead of waiting for it
            raise ShardError("worker rank(s) %s are gone (exit code(s) %s)" % ([r for r, _ in gone], [c for _, c in gone]))
        torch, dist = self._torch, self._dist
        ints = np.asarray(ints, dtype=np.int64).reshape(-1)
        floats = np.ascontiguousarray(floats, dtype=np.float64).reshape(-1)
        hdr = torch.tensor([op, ints.size, floats.size], dtype=torch.int64, device=self._cdev)
        self._coll(lambda: dist.broadcast(hdr, self.root, group=self.group))
        if ints.size + floats.size:
            payload = torch.from_numpy(np.concatenate([ints, floats.view(np.int64)])).to(self._cdev)
            self._coll(lambda: dist.broadcast(payload, self.root, group=self.group))

    def _agree(self, err: Optional[BaseException]):
        """Served mode: every rank reports whether its local step failed BEFORE the collective that would otherwise hang on the
        failed rank's peers; the root raises, the workers carry on serving."""
        if not self.served:
            if err is not None:
                raise err
            return
        torch, dist = self._torch, self._dist
        flags = np.zeros(self.world, dtype=np.int64)            # built on the host: a fill kernel on torch's null stream would not
        if err is not None:                                     # be ordered against the collective on the database's stream
            flags[self.rank] = 1
        bad = torch.from_numpy(flags).to(self._cdev)
        self._coll(lambda: dist.all_reduce(bad, op=dist.ReduceOp.SUM, group=self.group))
        failed = [r for r, b in enumerate(self._host(bad).tolist()) if b]
        if failed:
            raise ShardError("rank(s) %s of the sharded database failed%s" % (failed, ": %r" % (err,) if err is not None else "")) from err

    def _local_step(self, fn):
        """Run this rank's part; in served mode agree on success with the peers (see _agree)."""
        err, out = None, None
        try:
            out = fn()
        except Exception as e:          # noqa: BLE001 -- reported through _agree
            err = e
        self._agree(err)
        return out

    def _result_tensors(self, kinds):
        """This rank's avg [n,S] f64 / n_e [n,S] i32 / scores [n] f64 on the collective's device: zero-copy views of the
        library's own device memory under RCCL, host arrays otherwise."""
        torch = self._torch
        if self._cdev.type == "cuda" and hasattr(self.local, "device_tensor"):
            return [self.local.device_tensor(k) for k in kinds]
        if kinds == ["scores"]:
            return [torch.from_numpy(np.ascontiguousarray(self.local.scores()))]
        avg, ne = self.local.similarities()
        return [torch.from_numpy(np.ascontiguousarray({"avg": avg, "ne": ne}[k])) for k in kinds]

    def _gather(self, kinds):
        parts = self._local_step(lambda: self._result_tensors(kinds))
        return [self._host(self._coll(lambda p=p: all_gather_rows(p, self.n, self.group))) for p in parts]

    def _sum(self, arr: np.ndarray) -> np.ndarray:
        """All-reduce SUM of an array in which every element is non-zero on at most one rank (x + 0 is exact)."""
        t = self._torch.from_numpy(np.ascontiguousarray(arr)).to(self._cdev)
        self._coll(lambda: self._dist.all_reduce(t, op=self._dist.ReduceOp.SUM, group=self.group))
        return self._host(t)

    def _mine(self, rows_global: np.ndarray):
        """(positions in the list, local row numbers) of the global rows this rank holds."""
        rows_global = np.asarray(rows_global, dtype=np.int64).reshape(-1)
        if rows_global.size and (rows_global.min() < 0 or rows_global.max() >= self.n):
            raise ValueError("row outside [0,%d)" % self.n)
        pos = np.flatnonzero((rows_global >= self.row0) & (rows_global < self.row0 + self.local.n))
        return pos, rows_global[pos] - self.row0

    # ------------------------------------------------------------------ query
    @_atomic
    def set_layout(self, layout: str):
        """Every rank re-tiles its own rows in place (FeatureDB.set_layout): once, after loading."""
        self._announce(OP_LAYOUT, ints=[1 if layout == "tiled" else 0])
        self._local_step(lambda: self.local.set_layout(layout))

    @_atomic
    def restrict_slots(self, slot_used):
        used = None if slot_used is None else np.asarray(slot_used, dtype=bool)
        if used is not None and used.shape != (self.S, self.E):
            raise ValueError("slot_used must be [S,E]")
        if used is not None and used.all():
            used = None
        before = getattr(self, "_slots_used", "unset")
        if not isinstance(before, str) and ((used is None and before is None) or (used is not None and before is not None and (used == before).all())):
            return                                           # the restriction the ranks already hold: nothing to announce
        self._slots_used = "unset"                           # until every rank has taken the new restriction
        self._announce(OP_RESTRICT, ints=[-1] if used is None else used.astype(np.int64).reshape(-1))
        self._local_step(lambda: self.local.restrict_slots(used))
        self._slots_used = None if used is None else used.copy()

    @_atomic
    def set_query(self, t):
        """SPMD: the root's ``t`` is broadcast (the other ranks may pass None).  Served: it travels with the announcement."""
        torch, dist = self._torch, self._dist
        if self.served:
            # validated BEFORE the announcement: a root that raised after it would leave the workers waiting in _agree for a
            # partner that never comes, and every later operation out of step
            t = np.ascontiguousarray(t, dtype=np.float64)
            if t.shape != (self.S, self.E, self.D):
                raise ValueError("query must be [%d,%d,%d]" % (self.S, self.E, self.D))
            self._announce(OP_SET_QUERY, floats=t)
        else:
            buf = torch.zeros((self.S, self.E, self.D), dtype=torch.float64)
            if self.rank == self.root:
                buf.copy_(torch.from_numpy(np.ascontiguousarray(t, dtype=np.float64).reshape(self.S, self.E, self.D)))
            buf = buf.to(self._cdev)
            if self.world > 1:
                self._coll(lambda: dist.broadcast(buf, self.root, group=self.group))
            t = self._host(buf)
        if np.shape(t) != (self.S, self.E, self.D):
            raise ValueError("query must be [%d,%d,%d]" % (self.S, self.E, self.D))
        self._local_step(lambda: self.local.set_query(t))

    @_atomic
    def set_query_from_row(self, row: int, want: bool = True):
        """t = r/(r.r) of a resident clip (target_clip.py:311-313), computed on the GPU of the rank that holds the row and
        handed to the others (80 KB)."""
        row = int(row)
        if not 0 <= row < self.n:
            raise ValueError("row %d outside [0,%d)" % (row, self.n))
        self._announce(OP_QUERY_FROM_ROW, ints=[row])
        owner = self.row0 <= row < self.row0 + self.local.n
        t = self._local_step(lambda: self.local.set_query_from_row(row - self.row0, want=True) if owner else None)
        t = self._sum(t if owner else np.zeros((self.S, self.E, self.D), dtype=np.float64))
        if not owner:
            self._local_step(lambda: self.local.set_query(t))
        elif self.served:
            self._agree(None)
        return t if want else None

    @_atomic
    def scan(self, weights=None, keep_sims: bool = False):
        """Every rank scans its rows: one launch of scan_kernel each, no collective."""
        w = None if weights is None else np.ascontiguousarray(weights, dtype=np.float64)
        if w is not None and w.shape != (self.S,):
            raise ValueError("one weight per stream")
        self._announce(OP_SCAN, ints=[0 if w is None else 1, 1 if keep_sims else 0], floats=[] if w is None else w)
        self._local_step(lambda: self.local.scan(weights=w, keep_sims=keep_sims))

    @_atomic
    def rescore(self, weights):
        w = np.ascontiguousarray(weights, dtype=np.float64)
        if w.shape != (self.S,):
            raise ValueError("one weight per stream")
        self._announce(OP_RESCORE, floats=w)
        self._local_step(lambda: self.local.rescore(w))

    @_atomic
    def write_avg(self, avg, n_e=None):
        """Put averaged similarities [N,S] (and split counts) back on the ranks -- every rank takes its own rows.  How a ticket whose
        similarities were displaced by another ticket's round on the same database restores them before it scores (ticket.py)."""
        a = np.ascontiguousarray(avg, dtype=np.float64)
        ne = None if n_e is None else np.ascontiguousarray(n_e, dtype=np.int64)
        if a.shape != (self.n, self.S) or (ne is not None and ne.shape != (self.n, self.S)):
            raise ValueError("avg / n_e must be [%d,%d]" % (self.n, self.S))
        self._announce(OP_WRITE_AVG, ints=[] if ne is None else ne.reshape(-1), floats=a.reshape(-1))
        lo, hi = self.row0, self.row0 + self.local.n
        self._local_step(lambda: self.local.write_avg(a[lo:hi], None if ne is None else ne[lo:hi].astype(np.int32)))

    @_atomic
    def scan_batch(self, targets, weights, want: bool = True):
        """Q <= 16 queries in one pass of every rank over its rows -> scores [Q,N] (score slices gathered per query)."""
        t = np.ascontiguousarray(targets, dtype=np.float64)
        w = np.ascontiguousarray(weights, dtype=np.float64)
        if t.ndim != 4 or t.shape[1:] != (self.S, self.E, self.D) or w.shape != (t.shape[0], self.S):
            raise ValueError("targets must be [Q,%d,%d,%d] and weights [Q,%d]" % (self.S, self.E, self.D, self.S))
        self._announce(OP_SCAN_BATCH, ints=[t.shape[0], 1 if want else 0], floats=np.concatenate([t.reshape(-1), w.reshape(-1)]))
        out = self._local_step(lambda: self.local.scan_batch(t, w, want=want))
        if not want:
            return None
        loc = self._torch.from_numpy(np.ascontiguousarray(out.T)).to(self._cdev)              # [n_local, Q]
        full = self._coll(lambda: all_gather_rows(loc, self.n, self.group))
        return np.ascontiguousarray(self._host(full).T)

    # ------------------------------------------------------------------ results
    @_atomic
    def similarities(self, sims: bool = False):
        """(avg [N,S], n_e [N,S]) in global order; with ``sims`` also the per-split dots [N,S,E] (kept by ``scan(keep_sims=True)``)."""
        self._announce(OP_SIMS, ints=[1 if sims else 0])
        avg, ne = self._gather(["avg", "ne"])
        if not sims:
            return avg, ne
        part = self._local_step(lambda: self._torch.from_numpy(np.ascontiguousarray(self.local.similarities(sims=True)[2])).to(self._cdev))
        full = self._coll(lambda: all_gather_rows(part, self.n, self.group))
        return avg, ne, self._host(full)

    @_atomic
    def scores(self) -> np.ndarray:
        return self._host(self.scores_tensor())

    @_atomic
    def scores_tensor(self):
        """scores [N] in global order as a torch tensor on the collective's device (under RCCL: in HBM, no host copy -- what a
        caller that keeps working on the GPU, or a throughput measurement, wants)."""
        self._announce(OP_SCORES)
        part = self._local_step(lambda: self._result_tensors(["scores"]))[0]
        return self._coll(lambda: all_gather_rows(part, self.n, self.group))

    @property
    def stream(self):
        """The torch stream scans and collectives are ordered on (None on a host-memory backend)."""
        return self._stream

    @_atomic
    def scores_grid(self, w_grid, rows) -> np.ndarray:
        wg = np.ascontiguousarray(w_grid, dtype=np.float64)
        r = np.ascontiguousarray(rows, dtype=np.int64).reshape(-1)
        if wg.ndim != 2 or wg.shape[1] != self.S:
            raise ValueError("w_grid must be [G,%d]" % self.S)
        self._announce(OP_GRID, ints=np.concatenate([[wg.shape[0]], r]), floats=wg)

        def part():
            pos, mine = self._mine(r)
            out = np.zeros((wg.shape[0], r.size), dtype=np.float64)
            if pos.size:
                out[:, pos] = self.local.scores_grid(wg, mine)
            return out
        return self._sum(self._local_step(part))

    @_atomic
    def select(self, threshold: float, lower: float):
        """(match rows, near rows, first arg-max of the near rows) in GLOBAL row numbers, order preserved: each rank partitions
        its own slice on its GPU, the lists are concatenated rank-major (ticket.py:325-340)."""
        torch = self._torch
        self._announce(OP_SELECT, floats=[threshold, lower])

        def part():
            m, r, am = self.local.select(float(threshold), float(lower))
            best = float(self.local.scores_at([am])[0]) if am >= 0 else 0.0
            return m + self.row0, r + self.row0, (am + self.row0 if am >= 0 else -1), best
        return self._merge_selection(*self._local_step(part))

    def _merge_selection(self, m, r, am, best):
        """This rank's (match rows, near rows, first arg-max row or -1, its score) in GLOBAL row numbers -> the lists of all ranks,
        rank-major = database order, and the first arg-max over the ranks (two small gathers: sizes, then bodies)."""
        torch = self._torch
        head = torch.from_numpy(np.concatenate([np.array([m.size, r.size, am], dtype=np.int64),
                                                np.array([best], dtype=np.float64).view(np.int64)])).to(self._cdev)
        heads = self._host(self._coll(lambda: all_gather_rows(head.reshape(1, 4), self.world, self.group)))
        width = int((heads[:, 0] + heads[:, 1]).max())
        body = np.zeros(max(width, 1), dtype=np.int64)
        body[:m.size], body[m.size:m.size + r.size] = m, r
        bodies = self._host(self._coll(lambda: all_gather_rows(torch.from_numpy(body).reshape(1, -1).to(self._cdev), self.world, self.group)))
        match = np.concatenate([bodies[g, :heads[g, 0]] for g in range(self.world)])
        near = np.concatenate([bodies[g, heads[g, 0]:heads[g, 0] + heads[g, 1]] for g in range(self.world)])
        near_argmax, top = -1, None
        for g in range(self.world):                                   # first maximum in global order: lowest rank wins a tie
            if heads[g, 2] >= 0:
                v = float(heads[g, 3:4].view(np.float64)[0])
                if top is None or v > top:
                    near_argmax, top = int(heads[g, 2]), v
        return match, near, near_argmax

    @_atomic
    def query_round(self, t, weights=None, select=None):
        """ticket.py:120-180,311-356 -- similarities, scores and review partition of one query -- as ONE operation of the sharded
        database: one announcement (the query, the weights and the band travel with it), every rank's own vq_db_query_round, ONE
        gather of the per-row results ([n, 2S+1] doubles: avg | n_e | score) and the two small gathers of the selection -- where the
        separate calls (restrict, set_query, scan, similarities, rescore, scores, select) are seven announcements and six gathers.
        Same arrays, bit for bit (tests/test_sharded_db_gloo.py, tests/test_sharded_db_gpu.py).  SPMD: every rank passes the same
        arguments."""
        from .feature_db import RoundResult
        torch = self._torch
        S = self.S
        tq = None if t is None else np.ascontiguousarray(t, dtype=np.float64)
        w = None if weights is None else np.ascontiguousarray(weights, dtype=np.float64)
        if tq is not None and tq.shape != (S, self.E, self.D):
            raise ValueError("query must be [%d,%d,%d]" % (S, self.E, self.D))
        if w is not None and w.shape != (S,):
            raise ValueError("one weight per stream")
        if select is not None and w is None:
            raise ValueError("a selection needs the scores of the same call")
        if tq is None and w is None:
            raise ValueError("nothing to do")
        band = None if select is None else (float(select[0]), float(select[1]))
        flags = (1 if tq is not None else 0) | (2 if w is not None else 0) | (4 if band is not None else 0)
        self._announce(OP_ROUND, ints=[flags], floats=np.concatenate([np.zeros(0) if tq is None else tq.reshape(-1), np.zeros(0) if w is None else w,
                                                                        np.zeros(0) if band is None else np.array(band)]))

        def part():
            if hasattr(self.local, "query_round"):
                r = self.local.query_round(tq, weights=w, select=band)
            else:                                                     # a local database without the one-call form: its separate calls
                r = RoundResult()
                r.avg = r.n_e = r.scores = r.match_rows = r.near_rows = None
                r.near_argmax = -1
                if tq is not None:
                    self.local.set_query(tq)
                    self.local.scan(weights=w)
                    r.avg, r.n_e = self.local.similarities()
                elif w is not None:
                    self.local.rescore(w)
                if w is not None:
                    r.scores = self.local.scores()
                if band is not None:
                    r.match_rows, r.near_rows, r.near_argmax = self.local.select(*band)
            cols = ([r.avg, r.n_e.astype(np.float64)] if tq is not None else []) + ([r.scores[:, None]] if w is not None else [])
            best = float(r.scores[r.near_argmax]) if band is not None and r.near_argmax >= 0 else 0.0
            return r, np.ascontiguousarray(np.concatenate(cols, axis=1)), best
        r, packed, best = self._local_step(part)
        full = self._host(self._coll(lambda: all_gather_rows(torch.from_numpy(packed).to(self._cdev), self.n, self.group)))
        out = RoundResult()
        out.avg = out.n_e = out.scores = out.match_rows = out.near_rows = None
        out.near_argmax = -1
        c = 0
        if tq is not None:
            out.avg = np.ascontiguousarray(full[:, :S])
            out.n_e = full[:, S:2 * S].astype(np.int32)               # counts <= E: exact as doubles
            c = 2 * S
        if w is not None:
            out.scores = np.ascontiguousarray(full[:, c])
        if band is not None:
            am = int(r.near_argmax)
            out.match_rows, out.near_rows, out.near_argmax = self._merge_selection(
                np.asarray(r.match_rows, dtype=np.int64) + self.row0, np.asarray(r.near_rows, dtype=np.int64) + self.row0,
                am + self.row0 if am >= 0 else -1, best)
        return out

    @_atomic
    def topk(self, k: int):
        torch = self._torch
        k = int(min(k, self.n))
        self._announce(OP_TOPK, ints=[k])

        def part():
            rows, vals = self.local.topk(min(k, self.local.n)) if self.local.n else (np.zeros(0, np.int64), np.zeros(0))
            buf = np.zeros((1, 1 + 2 * k), dtype=np.int64)
            buf[0, 0] = rows.size
            buf[0, 1:1 + rows.size] = rows
            buf[0, 1 + k:1 + k + rows.size] = np.ascontiguousarray(vals, dtype=np.float64).view(np.int64)
            return buf
        buf = self._local_step(part)
        allb = self._host(self._coll(lambda: all_gather_rows(torch.from_numpy(buf).to(self._cdev), self.world, self.group)))
        rows = [allb[g, 1:1 + allb[g, 0]] for g in range(self.world)]
        vals = [allb[g, 1 + k:1 + k + allb[g, 0]].view(np.float64) for g in range(self.world)]
        row0s = [shard_range(self.n, self.world, g)[0] for g in range(self.world)]
        return merge_topk(rows, vals, row0s, k)

    @_atomic
    def min_score(self, rows) -> float:
        r = np.ascontiguousarray(rows, dtype=np.int64).reshape(-1)
        self._announce(OP_MIN, ints=r)

        def part():
            _pos, mine = self._mine(r)
            return self.local.min_score(mine)              # 1 when the rank holds none of the rows (ticket.py:302)
        m = self._local_step(part)
        t = self._torch.tensor([1.0 if m is None else m], dtype=self._torch.float64, device=self._cdev)
        self._coll(lambda: self._dist.all_reduce(t, op=self._dist.ReduceOp.MIN, group=self.group))
        return float(self._host(t)[0])

    @_atomic
    def read_rows(self, rows) -> np.ndarray:
        """[L,S,E,D] feature rows by GLOBAL row number, from whichever ranks hold them (L x 40 KB; the only time features move)."""
        r = np.ascontiguousarray(rows, dtype=np.int64).reshape(-1)
        self._announce(OP_FETCH, ints=r)

        def part():
            pos, mine = self._mine(r)
            out = np.zeros((r.size, self.S, self.E, self.D), dtype=self.dtype)
            if pos.size:
                out[pos] = self.local.read_rows(mine)
            return out
        return self._sum(self._local_step(part))

    @_atomic
    def bootstrap_target(self, valid_rows, invalid_rows=(), mu: float = 0.0, set_query: bool = True) -> np.ndarray:
        """New query vectors [S,E,D] from user-validated clips anywhere in the sharded database (target_clip.py:161-261): rows
        fetched from their owners, the closed forms on the root's GPU (same kernel as FeatureDB.bootstrap_target)."""
        from .bootstrap import bootstrap_targets
        v = np.ascontiguousarray(valid_rows, dtype=np.int64).reshape(-1)
        iv = np.ascontiguousarray(invalid_rows, dtype=np.int64).reshape(-1)
        if v.size < 1:
            raise ValueError("need at least one validated match")
        x = self.read_rows(np.concatenate([v, iv]))
        t = None
        if self.rank == self.root:
            problems = [(x[:v.size, s, e], x[v.size:, s, e] if iv.size else None) for s in range(self.S) for e in range(self.E)]
            t = bootstrap_targets(problems, mu, device=self.device).reshape(self.S, self.E, self.D)
        if self.served:                                    # only the root is here; the query travels with its announcement
            if set_query:
                self.set_query(t)
            return t
        t = self._sum(t if t is not None else np.zeros((self.S, self.E, self.D)))       # SPMD: every caller gets the vectors
        if set_query:
            self._local_step(lambda: self.local.set_query(t))
        return t

    # ------------------------------------------------------------------ index (global, on every rank)
    def row_of(self, clip_id: int) -> int:
        if self._row_of is None:
            self._row_of = {int(c): i for i, c in enumerate(self.clip_ids.tolist())}
        return self._row_of[int(clip_id)]

    def has_clip(self, clip_id) -> bool:
        if self._row_of is None:
            self.row_of(int(self.clip_ids[0]))
        try:
            return int(clip_id) in self._row_of
        except (TypeError, ValueError):
            return False

    # ------------------------------------------------------------------ the worker side
    def serve(self):
        """Ranks other than the root, served mode: execute what the root announces until it closes the database."""
        torch, dist = self._torch, self._dist
        if not self.served or self.rank == self.root:
            raise RuntimeError("serve() is for the non-root ranks of a served database")
        while True:
            hdr = self._coll(lambda: torch.zeros(3, dtype=torch.int64, device=self._cdev))      # filled on the collective's stream
            self._coll(lambda: dist.broadcast(hdr, self.root, group=self.group))
            op, ni, nf = (int(v) for v in self._host(hdr).tolist())
            ints, floats = np.zeros(0, np.int64), np.zeros(0)
            if ni + nf:
                payload = self._coll(lambda: torch.zeros(ni + nf, dtype=torch.int64, device=self._cdev))
                self._coll(lambda: dist.broadcast(payload, self.root, group=self.group))
                raw = self._host(payload)
                ints, floats = raw[:ni], raw[ni:].view(np.float64)
            if op == OP_CLOSE:
                break
            try:
                self._dispatch(op, ints, floats)
            except ShardError:
                pass                                        # agreed on and raised by the root; keep serving

    def _dispatch(self, op, ints, floats):
        S, E, D = self.S, self.E, self.D
        if op == OP_RESTRICT:
            self.restrict_slots(None if ints[0] < 0 else ints.reshape(S, E).astype(bool))
        elif op == OP_SET_QUERY:
            self._local_step(lambda: self.local.set_query(floats.reshape(S, E, D)))
        elif op == OP_QUERY_FROM_ROW:
            self.set_query_from_row(int(ints[0]), want=False)
        elif op == OP_SCAN:
            self.scan(weights=floats if ints[0] else None, keep_sims=bool(ints[1]))
        elif op == OP_RESCORE:
            self.rescore(floats)
        elif op == OP_SCAN_BATCH:
            q = int(ints[0])
            self.scan_batch(floats[:q * S * E * D].reshape(q, S, E, D), floats[q * S * E * D:].reshape(q, S), want=bool(ints[1]))
        elif op == OP_SIMS:
            self.similarities(sims=bool(ints[0]))
        elif op == OP_SCORES:
            self.scores()
        elif op == OP_GRID:
            self.scores_grid(floats.reshape(int(ints[0]), S), ints[1:])
        elif op == OP_SELECT:
            self.select(float(floats[0]), float(floats[1]))
        elif op == OP_ROUND:
            flags, k = int(ints[0]), 0
            tq = w = band = None
            if flags & 1:
                tq, k = floats[:S * E * D].reshape(S, E, D), S * E * D
            if flags & 2:
                w, k = floats[k:k + S], k + S
            if flags & 4:
                band = (float(floats[k]), float(floats[k + 1]))
            self.query_round(tq, weights=w, select=band)
        elif op == OP_TOPK:
            self.topk(int(ints[0]))
        elif op == OP_MIN:
            self.min_score(ints)
        elif op == OP_FETCH:
            self.read_rows(ints)
        elif op == OP_LAYOUT:
            self.set_layout("tiled" if ints[0] else "rows")
        elif op == OP_WRITE_AVG:
            self.write_avg(floats.reshape(self.n, S), ints.reshape(self.n, S) if ints.size else None)
        else:
            raise RuntimeError("unknown operation %d announced" % op)

    # ------------------------------------------------------------------ teardown
    @_atomic
    def close(self):
        if self._closed:
            return
        self._closed = True
        try:
            if not any(p.poll() is not None for p in getattr(self, "_workers", ())):
                self._announce(OP_CLOSE)                 # (with a dead worker the broadcast would never complete)
        finally:
            self.local.close()
            if getattr(self, "_owns_group", False):
                from . import fanout
                try:
                    self._dist.destroy_process_group()
                finally:
                    fanout.wait_children(getattr(self, "_workers", []), timeout_s=60.0)

    def __del__(self):
        try:
            if not self._closed and not self.served:
                self.local.close()
        except Exception:       # noqa: BLE001
            pass


def _await_workers(store, keys, procs, timeout_s: float, what: str):
    """Wait until every key is in the rendezvous store, watching the worker processes meanwhile."""
    import time
    deadline = time.monotonic() + timeout_s
    while keys and not store.check(list(keys)):
        dead = [(i + 1, p.returncode) for i, p in enumerate(procs) if p.poll() is not None]
        if dead:
            raise ShardError("worker rank(s) %s exited with code(s) %s before they could %s (their messages are on stderr)"
                             % ([r for r, _ in dead], [c for _, c in dead], what))
        if time.monotonic() > deadline:
            raise ShardError("the workers did not %s within %.0f s" % (what, timeout_s))
        time.sleep(0.02)


def worker_main(argv=None) -> int:
    """``shard_worker.py``: one non-root rank of a served database (started by ShardedFeatureDB.open)."""
    import argparse
    import datetime
    ap = argparse.ArgumentParser()
    ap.add_argument("store")
    ap.add_argument("--backend", default="nccl")
    ap.add_argument("--timeout", type=float, default=600.0)
    args = ap.parse_args(argv)
    import torch
    import torch.distributed as dist
    from .feature_db import FeatureDB
    from .feature_store import open_store
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    device = int(os.environ.get("VQ_FANOUT_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    _meta, feats, ids, _present = open_store(args.store)                  # a bad path fails HERE, before anybody waits in a collective
    row0, rows = shard_range(feats.shape[0], world, rank)
    timeout = datetime.timedelta(seconds=args.timeout)
    store = dist.TCPStore(os.environ.get("MASTER_ADDR", "127.0.0.1"), int(os.environ["MASTER_PORT"]), world, False, timeout)
    store.set("ready/%d" % rank, "1")
    kw = {"rank": rank, "world_size": world, "store": store, "timeout": timeout}
    if args.backend == "nccl":
        torch.cuda.set_device(device)
        kw["device_id"] = torch.device("cuda", device)
    dist.init_process_group(args.backend, **kw)
    local = FeatureDB.from_store(args.store, device=device, row0=row0, rows=rows)
    store.set("loaded/%d" % rank, "1")
    db = ShardedFeatureDB(local, feats.shape[0], row0, ids, served=True)
    try:
        db.serve()
    finally:
        db.local.close()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(worker_main())
