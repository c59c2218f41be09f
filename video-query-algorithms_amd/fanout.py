"""GPU fan-out of the reference's own command lines: one FRESH child process per GPU.

The reference maps its worker processes to GPUs itself: ``calcSig_wOF.py --num_worker W --gpus g0 .. gk`` gives worker i
(1-based) the GPU ``gpu_list[(i - 1) % len(gpu_list)]`` (src/features_GPU_compute/calcSig_wOF.py:44-56, pool at :204-210; the
ensemble script spells ``--num_worker 24 --gpus 0 1 2 3 4 5 6 7``, calcSig_wOF_ensemble.sh:13-19), and
``build_wof_clips.py --num_gpu N --starting_gpu S`` gives worker i the device ``(i - 1) % N + S``
(src/features_GPU_compute/build_wof_clips.py:66).  The drop-ins keep one process per GPU: started by plain ``python`` with
more than one GPU named, the command line starts itself again once per GPU -- as children of a parent that has made NO GPU
call (a process that has touched the GPU must never be replaced or forked into a GPU user) -- waits for them and returns the
first non-zero exit code.  Under ``torchrun`` (RANK in the environment) nothing is started: the launcher already did.
"""
from __future__ import annotations

import os
import socket
import subprocess
import sys
import time
from typing import Dict, List, Optional, Sequence

CHILD_ENV = "VQ_FANOUT_CHILD"          # set in every child: a child never fans out again


def worker_devices(gpu_list: Optional[Sequence[int]], num_worker: int, visible: Optional[int] = None) -> List[int]:
    """The distinct GPUs the reference's ``num_worker`` workers would land on, in first-use order (calcSig_wOF.py:47-55):
    with ``--gpus``: ``gpu_list[(i - 1) % len]`` for i = 1..num_worker; without: worker i -> GPU i - 1 (there the reference
    asserts i <= 10; here the range is clipped to the GPUs the node has when that is known)."""
    n = max(1, int(num_worker))
    if gpu_list:
        seen: List[int] = []
        for i in range(1, n + 1):
            g = int(gpu_list[(i - 1) % len(gpu_list)])
            if g not in seen:
                seen.append(g)
        return seen
    if visible is not None and visible > 0:
        n = min(n, visible)
    return list(range(n))


def visible_gpus() -> int:
    """Number of GPUs of the node WITHOUT initialising one (``torch.cuda.device_count()`` reads the topology only)."""
    try:
        import torch
        return int(torch.cuda.device_count())
    except Exception:
        return 0


def is_child() -> bool:
    return CHILD_ENV in os.environ


def free_port() -> int:
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def run_children(program: str, argv: Sequence[str], child_envs: Sequence[Dict[str, str]], poll_s: float = 0.05) -> int:
    """Start ``python program argv`` once per entry of ``child_envs`` (its variables on top of this process's), wait
    for all.  The first child to fail ends the others (a rank that dies would leave its peers waiting in a collective) and
    its exit code is returned; 0 when every child returned 0."""
    cmd = [sys.executable, os.path.abspath(program)] + list(argv)
    procs = []
    for extra in child_envs:
        env = dict(os.environ)
        env.update(extra)
        env[CHILD_ENV] = "1"
        procs.append(subprocess.Popen(cmd, env=env))
    rc = 0
    live = list(procs)
    kill_at = None
    while live:
        for p in list(live):
            r = p.poll()
            if r is None:
                continue
            live.remove(p)
            if r != 0 and rc == 0:
                rc = r if r > 0 else 128 - r            # a signal's negative code becomes the shell's 128 + signal
                for q in live:
                    q.terminate()
                kill_at = time.monotonic() + 10.0        # a terminated child that ignores SIGTERM
        if live:
            if kill_at is not None and time.monotonic() > kill_at:
                for q in live:
                    q.kill()
            time.sleep(poll_s)
    return rc


def rank_envs(world: int, threads_per_rank: Optional[int] = None) -> List[Dict[str, str]]:
    """torchrun's variables for ``world`` ranks of one node over 127.0.0.1."""
    port = str(free_port())
    out = []
    for r in range(world):
        e = {"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(world), "LOCAL_WORLD_SIZE": str(world),
             "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": port}
        if threads_per_rank:
            e["OMP_NUM_THREADS"] = str(threads_per_rank)
        out.append(e)
    return out
