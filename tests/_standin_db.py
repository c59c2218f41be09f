"""Test infrastructure: the LOCAL surface of FeatureDB on the numpy oracle, for the CPU tests of the sharded B seam (there is no GPU
in the build container; what those tests check is the partition / announce / gather / merge logic of ShardedFeatureDB, with
the oracle standing in for every rank's kernels).  Never imported by the product."""
import numpy as np

import sim_oracle as so


class OracleFeatureDB:
    def __init__(self, feats, present=None, stream_names=None, slot_splits=None, fail_on_scan=False):
        self.x = np.ascontiguousarray(feats)
        self.n, self.S, self.E, self.D = self.x.shape
        self.dtype = self.x.dtype
        self.device = 0
        self.present = None if present is None or np.asarray(present).all() else np.asarray(present).astype(np.uint8)
        self.stream_names, self.slot_splits = stream_names, slot_splits
        self._hidden = None
        self._t = self._avg = self._ne = self._scores = None
        self.fail_on_scan = fail_on_scan
        self.closed = False

    def set_stream(self, _s):
        pass

    def set_layout(self, layout):
        self.layout = layout

    def restrict_slots(self, slot_used):
        used = None if slot_used is None else np.asarray(slot_used, dtype=bool)
        self._hidden = None if used is None or used.all() else ~used

    def set_query(self, t):
        self._t = np.array(t, dtype=np.float64).reshape(self.S, self.E, self.D)
        self._avg = self._scores = None

    def set_query_from_row(self, row, want=True):
        self.set_query(np.stack([[so.scale_feature(self.x[row, s, e].astype(np.float64)) for e in range(self.E)] for s in range(self.S)]))
        return self._t.copy() if want else None

    def _effective(self):
        base = np.ones((self.n, self.S, self.E), dtype=bool) if self.present is None else self.present.astype(bool)
        return base if self._hidden is None else base & ~self._hidden[None]

    def scan(self, weights=None, keep_sims=False):
        if self.fail_on_scan:
            raise RuntimeError("stand-in: this rank's scan fails")
        self._sims, self._avg, self._ne = so.dense_similarities(self.x, self._t, self._effective())
        self._scores = None if weights is None else so.dense_scores(self._avg, weights)

    def scan_batch(self, targets, weights, want=True):
        out = np.empty((len(targets), self.n))
        for q, (t, w) in enumerate(zip(targets, weights)):
            _, avg, _ = so.dense_similarities(self.x, t, self._effective())
            out[q] = so.dense_scores(avg, w)
        return out if want else None

    def similarities(self, sims=False):
        if self._avg is None:
            raise RuntimeError("no similarities cached")
        return (self._avg.copy(), self._ne.copy(), self._sims.copy()) if sims else (self._avg.copy(), self._ne.copy())

    def write_avg(self, avg, n_e=None):
        self._avg = np.array(avg, dtype=np.float64)
        if n_e is not None:
            self._ne = np.array(n_e, dtype=np.int32)
        self._scores = None

    def rescore(self, weights):
        self._scores = so.dense_scores(self._avg, weights)

    def scores(self):
        if self._scores is None:
            raise RuntimeError("no scores computed")
        return self._scores.copy()

    def scores_at(self, rows):
        return self._scores[np.asarray(rows, dtype=np.int64)]

    def scores_grid(self, w_grid, rows):
        rows = np.asarray(rows, dtype=np.int64)
        return np.stack([so.dense_scores(self._avg[rows], w) for w in np.asarray(w_grid)])

    def select(self, threshold, lower):
        v = self._scores
        match = np.flatnonzero(v >= threshold)
        near = np.flatnonzero((lower <= v) & (v < threshold))
        return match, near, (int(near[np.argmax(v[near])]) if near.size else -1)

    def topk(self, k):
        return so.dense_topk(self._scores, k)

    def min_score(self, rows):
        m = 1.0
        for r in np.asarray(rows, dtype=np.int64):
            m = min(m, self._scores[r])
        return m

    def read_rows(self, rows):
        return self.x[np.asarray(rows, dtype=np.int64)]

    def close(self):
        self.closed = True
