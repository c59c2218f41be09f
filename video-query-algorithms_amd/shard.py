"""Multi-GPU plumbing: one process per GPU, ``torch.distributed`` (backend "nccl" = RCCL over xGMI on the
MI355X node, "gloo" in the CPU tests).  Both halves of the hot path shard over independent units:

* feature extraction: clips -- rank g extracts the contiguous clip range ``shard_range(n_clips, world, g)``
  (the reference's clip-level data parallelism, calcSig_wOF.py:204-210) and the per-GPU feature blocks are
  all-gathered so every rank holds ``[n_clips, D]`` in global clip order;
* scoring: database rows -- the DB is row-sharded, each rank scans its rows, and the score slices
  (N x 8 bytes, not the 41 GB of features) are all-gathered; selection then runs on one rank so that
  ``random.sample`` keeps CPython's semantics (ticket.py:333,341).

Nothing here computes: it only partitions and exchanges.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np


def shard_range(n: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous, balanced partition: (first unit, number of units) of `rank`; earlier ranks take the remainder."""
    base, rem = divmod(int(n), int(world))
    return rank * base + min(rank, rem), base + (1 if rank < rem else 0)


def all_gather_rows(local, n_total: int, group=None):
    """All-gather row blocks of unequal length (shard_range partition) into ``[n_total, ...]`` in global order.

    ``local`` is a torch tensor ``[rows_of_this_rank, ...]`` on the backend's device.  Blocks are padded to the
    largest shard so one fixed-size collective (``all_gather_into_tensor``) moves everything.
    """
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    if world == 1:
        return local
    if local.is_cuda and dist.get_backend(group) == "gloo":      # rehearsal of the N > 1 path without RCCL: stage through the host
        return all_gather_rows(local.cpu(), n_total, group).to(local.device)
    max_rows = -(-n_total // world)
    tail = tuple(local.shape[1:])
    padded = local
    if local.shape[0] != max_rows:
        padded = torch.zeros((max_rows,) + tail, dtype=local.dtype, device=local.device)
        padded[:local.shape[0]].copy_(local)
    out = torch.empty((world * max_rows,) + tail, dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, padded.contiguous(), group=group)       # rank-major = clip-major
    if n_total % world == 0:
        return out
    parts = [out[r * max_rows:r * max_rows + shard_range(n_total, world, r)[1]] for r in range(world)]
    return torch.cat(parts, dim=0)


def merge_topk(rows_per_rank: Sequence[np.ndarray], vals_per_rank: Sequence[np.ndarray], row0_per_rank: Sequence[int],
               k: int):
    """Merge per-shard top-k lists (local row indices) into the global top-k: descending score, ties by ascending
    GLOBAL row -- the order a stable sort of the whole score array would give (ticket.py:266)."""
    rows = np.concatenate([np.asarray(r, dtype=np.int64) + int(r0) for r, r0 in zip(rows_per_rank, row0_per_rank)])
    vals = np.concatenate([np.asarray(v, dtype=np.float64) for v in vals_per_rank])
    order = np.lexsort((rows, -vals))[:k]
    return rows[order], vals[order]
