"""The tunables of a query and their re-fit from the user's verdicts: the seam of the reference's ``Hyperparameter``
(src/models/hyperparameter.py) over the resident database.

The constructor signature, the attribute names and the two grids are the seam (broker.py:36-59 builds the object,
``compute_matches`` and ``TargetClip`` read its fields).  ``optimize_weights(ticket)`` keeps the contract of
hyperparameter.py:29-76 -- it sets ``weights`` and ``threshold`` and leaves ``ticket.scores`` at the last grid
weight -- but works on arrays: the 40 full rescoring passes of :57-58 become one launch of the grid kernel over the
labelled rows only (the loss of :60-64 reads nothing else), the 40 x 31 loss surface is accumulated clip by clip in
the reference's order so every cell sees the same sequence of fp64 additions, and the sub-grid refinement (:78-114)
is the separable three-point parabola below.
"""
from __future__ import annotations

import contextlib
import logging
import os

import numpy as np

FIT_TOLERANCE = 1e-6          # summed misfit above which the refinement is distrusted (hyperparameter.py:99)


def parabola_through(xs, y_lo, y_mid, y_hi):
    """Vertex and curvature of the parabola through (xs[0], y_lo), (xs[1], y_mid), (xs[2], y_hi) -- one axis of the
    separable surface a (w - w0)^2 + b (th - th0)^2 + c of hyperparameter.py:85-93 (same operation order)."""
    lo, mid, hi = xs
    rise, fall_hi, fall_lo = y_hi - y_lo, y_mid - y_hi, y_mid - y_lo
    vertex = 0.5 * (rise * mid ** 2 + fall_hi * lo ** 2 - fall_lo * hi ** 2) / (rise * mid + fall_hi * lo - fall_lo * hi)
    return vertex, fall_lo / ((mid - vertex) ** 2 - (lo - vertex) ** 2)


class Hyperparameter:
    def __init__(self, default_weights, default_threshold=0.8, ballast=0.3, near_miss_default=0.5, mu=.3,
                 streams=('rgb', 'warped_optical_flow'), feature_name='global_pool', f_bootstrap=0.5, f_memory=0.5,
                 bootstrap_type='simple', nbags=3):
        self.streams, self.feature_name = streams, feature_name
        self.default_weights, self.default_threshold = default_weights, default_threshold
        self.weights, self.threshold = {}, default_threshold
        self.ballast, self.near_miss_default = ballast, near_miss_default
        self.mu, self.f_bootstrap, self.f_memory = mu, f_bootstrap, f_memory
        self.bootstrap_type, self.nbags = bootstrap_type, nbags
        self.weight_grid = np.arange(0.5, 2.5, 0.05)           # hyperparameter.py:20-21
        self.threshold_grid = np.arange(0.5, 1.1, 0.02)

    def optimize_weights(self, ticket):
        rows, labels = self._labelled_rows(ticket)
        db = ticket.feature_db
        first, second = (ticket._stream_names.index(st) for st in self.streams[:2])
        candidates = np.zeros((len(self.weight_grid), db.S))
        candidates[:, first], candidates[:, second] = 1.0, self.weight_grid
        on_device = hasattr(db, "loss_surface") and 0 < len(labels) <= 4096
        with getattr(db, "lock", None) or contextlib.nullcontext():
            if hasattr(ticket, "_own_similarities"):
                ticket._own_similarities()      # a resident database shared between tickets: this ticket's similarities (ticket.py)
            if on_device:
                # the 40 rescorings AND the 40 x 31 loss surface in one launch: per cell the reference's fp64 operations, the labelled clips
                # added one at a time in the dict's order (csrc/vq_sim.hip: loss_surface_kernel) -- what the numpy lines below do on the host
                surface = db.loss_surface(candidates, rows, [float(v) for v in labels], self.threshold_grid, self.ballast)
            else:
                graded = db.scores_grid(candidates, rows)                           # [40][L] in one launch
        if on_device:
            surface = surface / len(labels)
            return self._pick(ticket, surface)
        th = self.threshold_grid[None, :]
        surface = np.tile(0.5 * th, (len(self.weight_grid), 1))
        # every clip's term of hyperparameter.py:60-64 at once ([L][40][31], the same elementwise operations in the same order), then
        # added to the surface ONE CLIP AT A TIME in dict order: every cell sees the reference's sequence of fp64 additions
        y = np.array([float(v) for v in labels], dtype=np.float64)[:, None, None]
        margin = graded.T[:, :, None] - th[None]
        terms = (np.heaviside(margin, 1) - y) * margin * (1 + y * self.ballast)
        for term in terms:
            surface = surface + term
        surface = surface / len(labels)
        return self._pick(ticket, surface)

    def _pick(self, ticket, surface):
        """Weights and threshold from the loss surface (hyperparameter.py:66-76)."""
        iw, it = np.unravel_index(np.argmin(surface), surface.shape)                # first minimum, row-major
        ticket.compute_scores({self.streams[0]: 1.0, self.streams[1]: self.weight_grid[-1]})   # what the 40 passes leave
        on_rim = iw in (0, len(self.weight_grid) - 1) or it in (0, len(self.threshold_grid) - 1)
        w_best, th_best = (self.weight_grid[iw], self.threshold_grid[it]) if on_rim else self._refine(surface, iw, it)
        self.weights = {self.streams[0]: 1.0, self.streams[1]: w_best}
        self.threshold = th_best - float(os.environ["COMPUTE_EPS"])                 # hyperparameter.py:5,75

    @staticmethod
    def _labelled_rows(ticket):
        """(DB rows, labels) of the clips of the previous round: the user's verdict where there is one, the engine's own
        call otherwise (hyperparameter.py:45-50).  A clip listed twice keeps its first position and last label, like
        the dict of the reference; an unknown clip is a KeyError, like ``ticket.scores[clip]`` there."""
        label = {}
        for m in ticket.matches:
            verdict = m["user_match"]
            label[m["video_clip"]] = m["is_match"] if verdict is None else verdict
        return [ticket.feature_db.row_of(clip) for clip in label], list(label.values())

    def _refine(self, surface, iw, it):
        """Minimum of the separable parabola through the best cell and its four neighbours, clamped to the
        neighbouring grid lines; falls back to the cell itself when the five points are not reproduced
        (hyperparameter.py:78-114)."""
        ws, ths = self.weight_grid[iw - 1:iw + 2], self.threshold_grid[it - 1:it + 2]
        centre = surface[iw, it]
        west, east, south, north = surface[iw - 1, it], surface[iw + 1, it], surface[iw, it - 1], surface[iw, it + 1]
        w0, a = parabola_through(ws, west, centre, east)
        th0, b = parabola_through(ths, south, centre, north)
        c = centre - a * (ws[1] - w0) ** 2 - b * (ths[1] - th0) ** 2
        w0, th0 = max(min(w0, ws[2]), ws[0]), max(min(th0, ths[2]), ths[0])

        def model(w, t):
            return a * (w - w0) ** 2 + b * (t - th0) ** 2 + c
        misfit = (abs(west - model(ws[0], ths[1])) + abs(south - model(ws[1], ths[0])) + abs(centre - model(ws[1], ths[1]))
                  + abs(north - model(ws[1], ths[2])) + abs(east - model(ws[2], ths[1])))
        if misfit > FIT_TOLERANCE:
            logging.warning("hyperparameter quadratic fine tuning failed - resort to selecting optimum on grid "
                            "without further interpolation")
            return ws[1], ths[1]
        return w0, th0
