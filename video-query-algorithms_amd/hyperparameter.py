"""Drop-in for the reference's ``Hyperparameter`` (src/models/hyperparameter.py).

Same constructor, grids and ``optimize_weights(ticket)`` contract (hyperparameter.py:9-76).  The 40
full rescoring passes of the reference (hyperparameter.py:57-58) are one launch of the grid kernel
over the labelled clips only -- the loss (hyperparameter.py:60-64) reads nothing else -- and a last
rescoring at the final grid weight so that ``ticket.scores`` is left exactly as the reference
leaves it (at w = 2.45).
"""
from __future__ import annotations

import logging
import os

import numpy as np


class Hyperparameter:
    def __init__(self, default_weights, default_threshold=0.8, ballast=0.3, near_miss_default=0.5, mu=.3,
                 streams=('rgb', 'warped_optical_flow'), feature_name='global_pool', f_bootstrap=0.5, f_memory=0.5,
                 bootstrap_type='simple', nbags=3):
        # hyperparameter.py:9-26
        self.default_weights = default_weights
        self.weights = {}
        self.default_threshold = default_threshold
        self.threshold = self.default_threshold
        self.near_miss_default = near_miss_default
        self.streams = streams
        self.feature_name = feature_name
        self.ballast = ballast
        self.weight_grid = np.arange(0.5, 2.5, 0.05)
        self.threshold_grid = np.arange(0.5, 1.1, 0.02)
        self.mu = mu
        self.f_bootstrap = f_bootstrap
        self.f_memory = f_memory
        self.bootstrap_type = bootstrap_type
        self.nbags = nbags

    def optimize_weights(self, ticket):
        """hyperparameter.py:29-76."""
        eps_threshold = float(os.environ["COMPUTE_EPS"])          # hyperparameter.py:5
        match_status = {}
        for match in ticket.matches:                               # hyperparameter.py:45-50
            if match["user_match"] is not None:
                match_status[match['video_clip']] = match["user_match"]
            else:
                match_status[match['video_clip']] = match["is_match"]
        clips = list(match_status)
        labels = [match_status[c] for c in clips]
        db = ticket.feature_db
        rows = [db.row_of(c) for c in clips]                       # KeyError like ticket.scores[clip] would
        s0 = ticket._stream_names.index(self.streams[0])
        s1 = ticket._stream_names.index(self.streams[1])
        w_grid = np.zeros((self.weight_grid.shape[0], db.S), dtype=np.float64)
        w_grid[:, s0] = 1.0
        w_grid[:, s1] = self.weight_grid
        grid_scores = db.scores_grid(w_grid, rows)                 # [40, L] on the GPU
        # loss grid (hyperparameter.py:56-65): accumulate over the labelled clips in dict order so
        # that every entry sees the same sequence of fp64 additions as the reference's scalar loop
        th = self.threshold_grid[None, :]
        losses = np.broadcast_to(0.5 * th, (w_grid.shape[0], th.shape[1])).copy()
        for j, y in enumerate(labels):
            d = grid_scores[:, j][:, None] - th
            losses = losses + (np.heaviside(d, 1) - y) * d * (1 + y * self.ballast)
        losses = losses / len(match_status)
        iw0, ith0 = np.unravel_index(np.argmin(losses, axis=None), losses.shape)
        # leave ticket.scores at the last grid weight, like the reference's loop does
        ticket.compute_scores({self.streams[0]: 1.0, self.streams[1]: self.weight_grid[-1]})
        if iw0 == 0 or ith0 == 0 or iw0 == len(self.weight_grid) - 1 or ith0 == len(self.threshold_grid) - 1:
            weight_optimum = self.weight_grid[iw0]
            threshold_optimum = self.threshold_grid[ith0]
        else:
            weight_optimum, threshold_optimum = self.fine_tune(iw0, ith0, losses)
        self.threshold = threshold_optimum - eps_threshold
        self.weights = {self.streams[0]: 1.0, self.streams[1]: weight_optimum}

    def fine_tune(self, iw0, ith0, losses):
        """hyperparameter.py:78-83."""
        wg, tg = self.weight_grid, self.threshold_grid
        xrange = [(wg[iw0 - 1], wg[iw0], wg[iw0 + 1]), (tg[ith0 - 1], tg[ith0], tg[ith0 + 1])]
        ydata = [losses[iw0 - 1, ith0], losses[iw0, ith0 - 1], losses[iw0, ith0], losses[iw0, ith0 + 1],
                 losses[iw0 + 1, ith0]]
        return self._quad_fit(xrange, ydata)

    @staticmethod
    def _quad_fit(x, y):
        """hyperparameter.py:85-114: vertex of a0 (w - w0)^2 + b0 (th - th0)^2 + c0 through five points."""
        (wl, wc, wr), (tl, tc, tr) = x
        y_wl, y_tl, y_c, y_tr, y_wr = y

        def vertex(lo, mid, hi, y_lo, y_hi):
            num = (y_hi - y_lo) * mid ** 2 + (y_c - y_hi) * lo ** 2 - (y_c - y_lo) * hi ** 2
            v = 0.5 * num / ((y_hi - y_lo) * mid + (y_c - y_hi) * lo - (y_c - y_lo) * hi)
            curv = (y_c - y_lo) / ((mid - v) ** 2 - (lo - v) ** 2)
            return v, curv

        w0, a0 = vertex(wl, wc, wr, y_wl, y_wr)
        th0, b0 = vertex(tl, tc, tr, y_tl, y_tr)
        c0 = y_c - a0 * (wc - w0) ** 2 - b0 * (tc - th0) ** 2
        w0 = max(min(w0, wr), wl)          # round-off on flat data may leave the bracket
        th0 = max(min(th0, tr), tl)
        fit = [a0 * (wl - w0) ** 2 + b0 * (tc - th0) ** 2 + c0,
               a0 * (wc - w0) ** 2 + b0 * (tl - th0) ** 2 + c0,
               a0 * (wc - w0) ** 2 + b0 * (tc - th0) ** 2 + c0,
               a0 * (wc - w0) ** 2 + b0 * (tr - th0) ** 2 + c0,
               a0 * (wr - w0) ** 2 + b0 * (tc - th0) ** 2 + c0]
        err = abs(y[0] - fit[0]) + abs(y[1] - fit[1]) + abs(y[2] - fit[2]) + abs(y[3] - fit[3]) + abs(y[4] - fit[4])
        if err > 10 ** -6:
            logging.warning("hyperparameter quadratic fine tuning failed - resort to selecting optimum on grid "
                            "without further interpolation")
            w0, th0 = wc, tc
        return w0, th0
