"""Resident feature database on one MI355X: the host-side handle over ``vq_db_*``.

The reference keeps no database object: every query re-downloads all features of the search set
as JSON and regroups them into ``{stream: {split: {clip: list}}}``
(``Ticket._get_candidate_features``, src/models/ticket.py:358-382).  ``FeatureDB`` holds the same
information as one ``[N][S][E][D]`` block in HBM plus the clip-id index and the order in which the
reference would have met the clips (which fixes the iteration order of ``ticket.similarities`` and
with it what ``random.sample`` draws in ``select_clips_to_review``).
"""
from __future__ import annotations

import ctypes as C
import threading
from typing import Iterable, Mapping, Sequence

import numpy as np

from . import _lib
from ._lib import VQ_F32, VQ_F64, call


def _np_ptr(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


class _RoundBlock:
    """One page-locked host block of vq_db_query_round (include/vq_amd.h), exposed to numpy.  The arrays a round hands out are views
    of it; when the last of them is gone the block goes back to its database's pool (or is freed if the database is)."""

    def __init__(self, ptr: int, nbytes: int, pool: dict):
        self.ptr, self.nbytes, self._pool = ptr, nbytes, pool
        self.__array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 3}

    def bytes(self) -> np.ndarray:
        return np.asarray(self)                               # its .base is self: views of it keep the block out of the pool

    def __del__(self):
        pool = self._pool
        if not pool["closed"] and len(pool["free"]) < 4:
            pool["free"].append((self.ptr, self.nbytes))      # reborn as a fresh _RoundBlock by the next round
        else:
            try:
                _lib.load().vq_host_free(C.c_void_p(self.ptr))
            except Exception:
                pass


class RoundResult:
    """What one call of :meth:`FeatureDB.query_round` brought back (numpy views of one page-locked block)."""
    __slots__ = ("avg", "n_e", "scores", "match_rows", "near_rows", "near_argmax")


class FeatureDB:
    """N clips x S streams x E ensemble slots x D floats, resident on ``device``."""

    def __init__(self, n: int, n_streams: int, n_splits: int, dim: int = 1024, dtype=np.float32, device: int = 0,
                 clip_ids: Sequence[int] | None = None):
        self.n, self.S, self.E, self.D = int(n), int(n_streams), int(n_splits), int(dim)
        self.dtype = np.dtype(dtype)
        if self.dtype not in (np.dtype(np.float32), np.dtype(np.float64)):
            raise TypeError("FeatureDB dtype must be float32 or float64")
        self.device = int(device)
        self._h = C.c_void_p()
        call("vq_db_create", self.n, self.S, self.E, self.D, VQ_F64 if self.dtype == np.float64 else VQ_F32,
             self.device, C.byref(self._h))
        self.clip_ids = (np.arange(1, self.n + 1, dtype=np.int64) if clip_ids is None
                         else np.asarray(clip_ids, dtype=np.int64).copy())
        if self.clip_ids.shape != (self.n,):
            raise ValueError("clip_ids must have shape (%d,)" % self.n)
        self._row_of = None
        self.present = None
        self._keepalive = None
        # One resident database may serve several tickets.  The query, the averaged similarities and the scores
        # are state of the HANDLE, and a round is several calls (ticket.py:96-170): ``lock`` (re-entrant) makes a group of calls
        # atomic, ``sims_owner`` / ``scores_owner`` say whose similarities / scores the device holds right now, so a ticket that finds
        # another ticket's there puts its own back (TicketScoring._own_similarities) instead of scoring with a stranger's query.
        self.lock = threading.RLock()
        self.sims_owner = None
        self.scores_owner = None

    # ------------------------------------------------------------------ construction
    @classmethod
    def from_arrays(cls, feats: np.ndarray, clip_ids=None, present=None, device: int = 0) -> "FeatureDB":
        feats = np.asarray(feats)
        if feats.ndim != 4:
            raise ValueError("feats must be [N,S,E,D]")
        if feats.dtype not in (np.float32, np.float64):
            feats = feats.astype(np.float32)
        n, s, e, d = feats.shape
        db = cls(n, s, e, d, feats.dtype, device, clip_ids)
        db.upload(0, feats)
        if present is not None:
            db.set_present(present)
        return db

    @classmethod
    def from_records(cls, records: Iterable[Mapping], target_features: Mapping, streams: Sequence[str],
                     feature_name: str, dtype=np.float64, device: int = 0) -> "FeatureDB":
        """Regroup API records exactly like ticket.py:358-382 + the walk of ticket.py:146-160.

        Slot ``e`` of stream ``s`` is the e-th split of ``target_features[s]`` (dict order = the order
        ``np.dot`` results are summed in at ticket.py:156); clips are numbered in the order the
        reference first meets them (stream-major, split, record order).
        """
        splits = set()
        for st in target_features:
            splits.update(target_features[st].keys())
        cand = {st: {sp: {} for sp in splits} for st in streams}
        for tf in records:                                            # ticket.py:374-381
            st = tf["dnn_stream_id"]
            if st in streams and tf["name"] == feature_name and tf["dnn_stream_split"] in splits:
                cand[st][tf["dnn_stream_split"]][tf["video_clip_id"]] = tf["feature_vector"]
        stream_list = list(target_features.keys())                    # iteration order of ticket.py:146
        slot_splits = [list(target_features[st].keys()) for st in stream_list]
        n_slots = max((len(x) for x in slot_splits), default=0)
        order = {}
        for st, sps in zip(stream_list, slot_splits):                 # first-seen order, ticket.py:146-160
            for sp in sps:
                for clip in cand[st][sp]:
                    if clip not in order:
                        order[clip] = len(order)
        n = len(order)
        if n == 0 or n_slots == 0:
            raise ValueError("no candidate features match the target's streams/splits")
        dim = None
        for st, sps in zip(stream_list, slot_splits):
            for sp in sps:
                for vec in cand[st][sp].values():
                    dim = len(vec)
                    break
        feats = np.zeros((n, len(stream_list), n_slots, dim), dtype=dtype)
        present = np.zeros((n, len(stream_list), n_slots), dtype=np.uint8)
        for si, (st, sps) in enumerate(zip(stream_list, slot_splits)):
            for ei, sp in enumerate(sps):
                d = cand[st][sp]
                if not d:
                    continue
                rows = np.fromiter((order[c] for c in d), dtype=np.int64, count=len(d))
                feats[rows, si, ei] = np.asarray(list(d.values()), dtype=dtype)
                present[rows, si, ei] = 1
        db = cls(n, len(stream_list), n_slots, dim, dtype, device,
                 np.fromiter(order.keys(), dtype=np.int64, count=n))
        db.stream_names = stream_list
        db.slot_splits = slot_splits
        db.upload(0, feats)
        if not present.all():
            db.set_present(present)
        return db

    @classmethod
    def from_store(cls, path: str, device: int = 0, row0: int = 0, rows: int | None = None,
                   chunk_rows: int = 16384) -> "FeatureDB":
        """Rows [row0, row0+rows) of a binary feature store (feature_store.py) -> resident DB; the memory-mapped block
        is uploaded in chunks of ``chunk_rows`` clips, so host memory stays bounded at any database size."""
        from .feature_store import open_store
        meta, feats, ids, present = open_store(path)
        n_all = feats.shape[0]
        rows = n_all - row0 if rows is None else int(rows)
        if row0 < 0 or rows <= 0 or row0 + rows > n_all:
            raise ValueError("rows [%d,%d) outside the store's %d clips" % (row0, row0 + rows, n_all))
        db = cls(rows, feats.shape[1], feats.shape[2], feats.shape[3], feats.dtype, device, ids[row0:row0 + rows])
        for r in range(0, rows, chunk_rows):
            k = min(chunk_rows, rows - r)
            db.upload(r, np.ascontiguousarray(feats[row0 + r:row0 + r + k]))
        if present is not None:
            db.set_present(np.ascontiguousarray(present[row0:row0 + rows]))
        db.stream_names = list(meta["streams"])
        db.slot_splits = [list(meta["splits"])] * len(meta["streams"])
        return db

    @classmethod
    def synthetic(cls, n: int, n_streams: int, n_splits: int, dim: int = 1024, seed: int = 0,
                  scales: Sequence[float] = (4.0, 1.0), row0: int = 0, dtype=np.float32, device: int = 0,
                  clip_ids=None) -> "FeatureDB":
        """Rows generated on the device by the counter-based hash (oracle.sim_oracle.synth_features)."""
        db = cls(n, n_streams, n_splits, dim, dtype, device, clip_ids)
        sc = np.ascontiguousarray(np.asarray(scales, dtype=np.float32))
        if sc.shape != (n_streams,):
            raise ValueError("scales must have one entry per stream")
        call("vq_db_generate", db._h, C.c_uint64(seed), row0, sc.ctypes.data_as(C.POINTER(C.c_float)))
        return db

    # ------------------------------------------------------------------ data movement
    def upload(self, row0: int, feats: np.ndarray):
        a = np.ascontiguousarray(feats, dtype=self.dtype)
        if a.shape[1:] != (self.S, self.E, self.D):
            raise ValueError("rows must be [n,%d,%d,%d]" % (self.S, self.E, self.D))
        call("vq_db_upload", self._h, int(row0), a.shape[0], _np_ptr(a))

    def set_layout(self, layout: str):
        """"rows" ([N][S][E][D]) or "tiled" ([tile of 16 clips][S*E][D/4][clip][4]: the order in which every load of the scans
        takes whole lines; fp32, D = 1024, S <= 2, E <= 5).  In place, one sweep of the block: call once after loading."""
        call("vq_db_set_layout", self._h, {"rows": _lib.VQ_LAYOUT_ROWS, "tiled": _lib.VQ_LAYOUT_TILED}[layout])

    @property
    def layout(self) -> str:
        v = C.c_int32()
        call("vq_db_layout", self._h, C.byref(v))
        return "tiled" if v.value == _lib.VQ_LAYOUT_TILED else "rows"

    def adopt_device(self, dev_ptr: int, keepalive=None):
        """Use caller-owned device memory (e.g. a torch tensor holding all-gathered blocks)."""
        call("vq_db_adopt_device", self._h, C.c_void_p(dev_ptr))
        self._keepalive = keepalive

    def set_present(self, present):
        """The database's OWN presence mask [N,S,E] (None = dense).  Per-query restrictions go through
        :meth:`restrict_slots` and never change it."""
        if present is None:
            self.present = None
        else:
            p = np.ascontiguousarray(np.asarray(present).astype(np.uint8))
            if p.shape != (self.n, self.S, self.E):
                raise ValueError("present must be [N,S,E]")
            self.present = p
        self._slots_hidden = None
        call("vq_db_set_present", self._h, None if self.present is None else _np_ptr(self.present))

    def restrict_slots(self, slot_used):
        """Hide the (stream, split) slots a query's target lacks (the reference never visits them, ticket.py:146-148)
        for the scans that follow; ``None`` / all-true lifts the restriction.  The mask on the device is rebuilt from
        the database's own mask every time, so one ragged query leaves nothing behind for the next."""
        used = None if slot_used is None else np.asarray(slot_used, dtype=bool)
        if used is not None and used.shape != (self.S, self.E):
            raise ValueError("slot_used must be [S,E]")
        if used is not None and used.all():
            used = None
        hidden = None if used is None else ~used
        before = getattr(self, "_slots_hidden", None)
        if (hidden is None and before is None) or (hidden is not None and before is not None and (hidden == before).all()):
            return
        if hidden is None:
            effective = self.present
        else:
            base = self.present if self.present is not None else np.ones((self.n, self.S, self.E), dtype=np.uint8)
            effective = np.ascontiguousarray(base * used[None].astype(np.uint8))
        call("vq_db_set_present", self._h, None if effective is None else _np_ptr(effective))
        self._slots_hidden = hidden

    def set_stream(self, hip_stream: int):
        call("vq_db_set_stream", self._h, C.c_void_p(hip_stream))

    def feats_devptr(self) -> int:
        p = C.c_void_p()
        call("vq_db_feats_devptr", self._h, C.byref(p))
        return p.value

    def scores_devptr(self) -> int:
        p = C.c_void_p()
        call("vq_db_scores_devptr", self._h, C.byref(p))
        return p.value

    def avg_devptr(self) -> int:
        p = C.c_void_p()
        call("vq_db_avg_devptr", self._h, C.byref(p))
        return p.value

    def device_tensor(self, kind: str):
        """Zero-copy torch view of a result array in the library's device memory -- "avg" [N,S] f64, "ne" [N,S] i32, "scores"
        [N] f64 -- for collectives that take device tensors (RCCL all-gather of score slices).  The caller orders its use
        behind the scan (same stream as ``set_stream``)."""
        _lib.require_torch_runtime("FeatureDB.device_tensor")
        import torch
        name, shape, typestr = {"avg": ("vq_db_avg_devptr", (self.n, self.S), "<f8"), "ne": ("vq_db_ne_devptr", (self.n, self.S), "<i4"),
                                "scores": ("vq_db_scores_devptr", (self.n,), "<f8")}[kind]
        p = C.c_void_p()
        call(name, self._h, C.byref(p))

        class _View:
            __cuda_array_interface__ = {"shape": shape, "typestr": typestr, "data": (int(p.value), False), "version": 2}
        return torch.as_tensor(_View(), device=torch.device("cuda", self.device))

    def read_rows(self, rows: Sequence[int]) -> np.ndarray:
        """Feature rows [L,S,E,D] back on the host (database dtype)."""
        r = np.ascontiguousarray(rows, dtype=np.int64).reshape(-1)
        out = np.empty((r.size, self.S, self.E, self.D), dtype=self.dtype)
        call("vq_db_read_rows", self._h, _np_ptr(r) if r.size else None, int(r.size), _np_ptr(out))
        return out

    def scores_at(self, rows: Sequence[int]) -> np.ndarray:
        r = np.ascontiguousarray(rows, dtype=np.int64).reshape(-1)
        out = np.empty(r.size, dtype=np.float64)
        call("vq_db_read_scores_at", self._h, _np_ptr(r) if r.size else None, int(r.size), _np_ptr(out))
        return out

    # ------------------------------------------------------------------ query
    def set_query(self, t: np.ndarray):
        t = np.ascontiguousarray(t, dtype=np.float64)
        if t.shape != (self.S, self.E, self.D):
            raise ValueError("query must be [%d,%d,%d]" % (self.S, self.E, self.D))
        call("vq_db_set_query", self._h, _np_ptr(t))

    def set_query_from_row(self, row: int, want: bool = True):
        """t = r/(r.r) of a resident row (target_clip.py:311-313), computed on the device."""
        out = np.empty((self.S, self.E, self.D), dtype=np.float64) if want else None
        call("vq_db_set_query_from_row", self._h, int(row), _np_ptr(out) if want else None)
        return out

    def bootstrap_target(self, valid_rows: Sequence[int], invalid_rows: Sequence[int] = (), mu: float = 0.0,
                         set_query: bool = True) -> np.ndarray:
        """New query vectors [S,E,D] from user-validated resident clips (target_clip.py:161-261, closed forms on the
        device; see csrc/vq_boot.hip).  With ``set_query`` they become the query of the next scan."""
        v = np.ascontiguousarray(valid_rows, dtype=np.int64)
        iv = np.ascontiguousarray(invalid_rows, dtype=np.int64)
        out = np.empty((self.S, self.E, self.D), dtype=np.float64)
        call("vq_db_bootstrap_target", self._h, _np_ptr(v), int(v.size), _np_ptr(iv) if iv.size else None, int(iv.size), float(mu),
             _np_ptr(out), 1 if set_query else 0)
        return out

    def scan(self, weights: Sequence[float] | None = None, keep_sims: bool = False):
        """One pass over the DB (ticket.py:120-163 [+ 165-180 when weights are given])."""
        w = None if weights is None else np.ascontiguousarray(weights, dtype=np.float64)
        if w is not None and w.shape != (self.S,):
            raise ValueError("one weight per stream")
        call("vq_db_scan", self._h, _np_ptr(w) if w is not None else None, 1 if keep_sims else 0)

    def scan_batch(self, targets: np.ndarray, weights: np.ndarray, want: bool = True):
        """Q <= 16 queries in one pass over the database: targets [Q,S,E,D] fp64, weights [Q,S] -> scores [Q,N].  The dots
        run on the fp64 matrix cores, so the scores equal Q single scans to rounding (<= 1e-12), not bit for bit; the
        single-query state of the DB is left alone."""
        t = np.ascontiguousarray(targets, dtype=np.float64)
        w = np.ascontiguousarray(weights, dtype=np.float64)
        if t.ndim != 4 or t.shape[1:] != (self.S, self.E, self.D) or w.shape != (t.shape[0], self.S):
            raise ValueError("targets must be [Q,%d,%d,%d] and weights [Q,%d]" % (self.S, self.E, self.D, self.S))
        out = np.empty((t.shape[0], self.n), dtype=np.float64) if want else None
        call("vq_db_scan_batch", self._h, t.shape[0], _np_ptr(t), _np_ptr(w), _np_ptr(out) if want else None)
        return out

    # ------------------------------------------------------------------ a query round in one call
    def _round_block(self) -> _RoundBlock:
        if getattr(self, "_round_off", None) is None:
            off = (C.c_int64 * 10)()
            call("vq_db_round_layout", self._h, off)
            self._round_off = [int(v) for v in off]
            self._round_pool = {"free": [], "closed": False}
        if self._round_pool["free"]:
            ptr, nbytes = self._round_pool["free"].pop()
        else:
            p = C.c_void_p()
            nbytes = self._round_off[8]
            call("vq_host_alloc", C.byref(p), nbytes)
            ptr = p.value
        return _RoundBlock(ptr, nbytes, self._round_pool)

    def query_round(self, t: np.ndarray | None, weights=None, select=None) -> RoundResult:
        """ticket.py:120-180,311-356 as ONE call of the library (vq_db_query_round): one lock, one synchronisation, one copy back.
        ``t`` [S,E,D]: scan under this query (None: the similarities the handle holds); ``weights`` [S]: scores under them;
        ``select`` = (threshold, lower): the order-preserving partition.  Bit for bit what set_query / scan / similarities / rescore /
        scores / select return one by one (tested); the arrays are views of page-locked memory that stay valid as long as they live."""
        blk = self._round_block()
        off = self._round_off
        raw = blk.bytes()
        n, S = self.n, self.S

        def view(piece, dtype, count, shape):
            return raw[off[piece]:off[piece] + count * dtype().itemsize].view(dtype).reshape(shape)
        flags = 0
        if t is not None:
            view(0, np.float64, S * self.E * self.D, (S, self.E, self.D))[...] = t
            flags |= 1
        if weights is not None:
            view(1, np.float64, S, (S,))[...] = weights
            flags |= 2
        th, lower = (float(select[0]), float(select[1])) if select is not None else (0.0, 0.0)
        if select is not None:
            flags |= 4
        call("vq_db_query_round", self._h, C.c_void_p(blk.ptr), blk.nbytes, flags, th, lower)
        r = RoundResult()
        r.avg = view(2, np.float64, n * S, (n, S)) if t is not None else None
        r.n_e = view(3, np.int32, n * S, (n, S)) if t is not None else None
        r.scores = view(4, np.float64, n, (n,)) if weights is not None else None
        r.match_rows = r.near_rows = None
        r.near_argmax = -1
        if select is not None:
            res = view(5, np.int64, 4, (4,))
            nm, nn, r.near_argmax = int(res[0]), int(res[1]), int(res[2])
            if nm <= off[9] and nn <= off[9]:
                r.match_rows = view(6, np.int64, nm, (nm,))
                r.near_rows = view(7, np.int64, nn, (nn,))
            else:                                             # a list longer than its prefix: the full lists are still on the device
                m = np.empty(nm, dtype=np.int64)
                q = np.empty(nn, dtype=np.int64)
                call("vq_db_select_fetch", self._h, _np_ptr(m) if nm else None, nm, _np_ptr(q) if nn else None, nn)
                r.match_rows, r.near_rows = m, q
        return r

    def rescore(self, weights: Sequence[float]):
        w = np.ascontiguousarray(weights, dtype=np.float64)
        if w.shape != (self.S,):
            raise ValueError("one weight per stream")
        call("vq_db_rescore", self._h, _np_ptr(w))

    def similarities(self, sims: bool = False):
        avg = np.empty((self.n, self.S), dtype=np.float64)
        ne = np.empty((self.n, self.S), dtype=np.int32)
        sm = np.empty((self.n, self.S, self.E), dtype=np.float64) if sims else None
        call("vq_db_read_similarities", self._h, _np_ptr(avg), _np_ptr(ne), _np_ptr(sm) if sims else None)
        return (avg, ne, sm) if sims else (avg, ne)

    def write_avg(self, avg: np.ndarray, n_e: np.ndarray | None = None):
        a = np.ascontiguousarray(avg, dtype=np.float64)
        ne = None if n_e is None else np.ascontiguousarray(n_e, dtype=np.int32)
        call("vq_db_write_avg", self._h, _np_ptr(a), _np_ptr(ne) if ne is not None else None)

    def scores(self) -> np.ndarray:
        out = np.empty(self.n, dtype=np.float64)
        call("vq_db_read_scores", self._h, _np_ptr(out))
        return out

    def scores_grid(self, w_grid: np.ndarray, rows: Sequence[int]) -> np.ndarray:
        wg = np.ascontiguousarray(w_grid, dtype=np.float64)
        r = np.ascontiguousarray(rows, dtype=np.int64)
        if wg.ndim != 2 or wg.shape[1] != self.S:
            raise ValueError("w_grid must be [G,%d]" % self.S)
        out = np.empty((wg.shape[0], r.shape[0]), dtype=np.float64)
        call("vq_db_scores_grid", self._h, _np_ptr(wg), wg.shape[0], _np_ptr(r), r.shape[0], _np_ptr(out))
        return out

    def loss_surface(self, w_grid: np.ndarray, rows: Sequence[int], labels: Sequence[float], th_grid: np.ndarray, ballast: float) -> np.ndarray:
        """hyperparameter.py:57-64 in one launch: [G, T] sums of the loss over the labelled rows (in order) for every grid weight and
        grid threshold, starting from 0.5 * threshold; the caller divides by the number of labels (include/vq_amd.h)."""
        wg = np.ascontiguousarray(w_grid, dtype=np.float64)
        r = np.ascontiguousarray(rows, dtype=np.int64).reshape(-1)
        y = np.ascontiguousarray(labels, dtype=np.float64).reshape(-1)
        th = np.ascontiguousarray(th_grid, dtype=np.float64).reshape(-1)
        if wg.ndim != 2 or wg.shape[1] != self.S or y.shape != r.shape:
            raise ValueError("w_grid must be [G,%d] and one label per row" % self.S)
        out = np.empty((wg.shape[0], th.size), dtype=np.float64)
        call("vq_db_loss_surface", self._h, _np_ptr(wg), wg.shape[0], _np_ptr(r), _np_ptr(y), r.size, _np_ptr(th), th.size, float(ballast), _np_ptr(out))
        return out

    def select(self, threshold: float, lower: float):
        """Order-preserving partition (ticket.py:325-340): (match_rows, near_rows, near_argmax).  Partition and
        copy-out are one locked call of the library, so handles shared between broker threads stay consistent."""
        nm, nn, am = C.c_int64(), C.c_int64(), C.c_int64()
        m = np.empty(self.n, dtype=np.int64)
        r = np.empty(self.n, dtype=np.int64)
        call("vq_db_select_rows", self._h, float(threshold), float(lower), _np_ptr(m), m.size, _np_ptr(r), r.size,
             C.byref(nm), C.byref(nn), C.byref(am))
        return m[:nm.value].copy(), r[:nn.value].copy(), am.value

    def topk(self, k: int):
        k = int(min(k, self.n))
        rows = np.empty(k, dtype=np.int64)
        vals = np.empty(k, dtype=np.float64)
        kk = C.c_int64()
        call("vq_db_topk", self._h, k, _np_ptr(rows), _np_ptr(vals), C.byref(kk))
        return rows[:kk.value], vals[:kk.value]

    def min_score(self, rows: Sequence[int]) -> float:
        r = np.ascontiguousarray(rows, dtype=np.int64)
        out = C.c_double()
        call("vq_db_min_score", self._h, _np_ptr(r) if r.size else None, r.size, C.byref(out))
        return out.value

    # ------------------------------------------------------------------ index
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

    def close(self):
        if self._h:
            pool = getattr(self, "_round_pool", None)
            if pool is not None:
                pool["closed"] = True                         # blocks still referenced by result arrays free themselves
                while pool["free"]:
                    _lib.load().vq_host_free(C.c_void_p(pool["free"].pop()[0]))
            _lib.load().vq_db_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
