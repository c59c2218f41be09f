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
    """Number of GPUs of the node, counted WITHOUT any HIP call (``torch.cuda.device_count()`` falls through to
    ``hipGetDeviceCount`` when amdsmi is not usable, and a parent that has initialised a GPU must not start GPU children):
    the KFD topology lists one node per agent, GPUs are the nodes with SIMDs; a ``*_VISIBLE_DEVICES`` list narrows it."""
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            return len([x for x in v.split(",") if x.strip() != ""])
    n = 0
    root = "/sys/class/kfd/kfd/topology/nodes"
    try:
        for node in os.listdir(root):
            try:
                with open(os.path.join(root, node, "properties")) as f:
                    props = dict(line.split()[:2] for line in f if len(line.split()) >= 2)
            except OSError:
                continue
            if int(props.get("simd_count", "0")) > 0:
                n += 1
    except OSError:
        return 0
    return n


def is_child() -> bool:
    return CHILD_ENV in os.environ


def free_port() -> int:
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def start_children(program: str, argv: Sequence[str], child_envs: Sequence[Dict[str, str]]) -> List[subprocess.Popen]:
    """Start ``python program argv`` once per entry of ``child_envs`` (its variables on top of this process's)."""
    cmd = [sys.executable, os.path.abspath(program)] + list(argv)
    procs: List[subprocess.Popen] = []
    try:
        for extra in child_envs:
            env = dict(os.environ)
            env.update(extra)
            env[CHILD_ENV] = "1"
            procs.append(subprocess.Popen(cmd, env=env))
    except BaseException:
        stop_children(procs)
        raise
    return procs


def stop_children(procs: Sequence[subprocess.Popen], grace_s: float = 10.0):
    """Terminate what is still running, kill what ignores it (a rank blocked in a collective does)."""
    live = [p for p in procs if p.poll() is None]
    for p in live:
        p.terminate()
    deadline = time.monotonic() + grace_s
    for p in live:
        try:
            p.wait(timeout=max(0.0, deadline - time.monotonic()))
        except subprocess.TimeoutExpired:
            p.kill()
            p.wait()


def wait_children(procs: Sequence[subprocess.Popen], timeout_s: float = 60.0) -> int:
    """Wait for children that were told to finish; whoever is still there after ``timeout_s`` is stopped.  First non-zero code."""
    rc = 0
    deadline = time.monotonic() + timeout_s
    for p in procs:
        try:
            r = p.wait(timeout=max(0.0, deadline - time.monotonic()))
        except subprocess.TimeoutExpired:
            stop_children([p])
            r = p.returncode
        if r and rc == 0:
            rc = r if r > 0 else 128 - r
    return rc


def run_children(program: str, argv: Sequence[str], child_envs: Sequence[Dict[str, str]], poll_s: float = 0.05) -> int:
    """Start ``python program argv`` once per entry of ``child_envs``, wait for all.  The first child to fail ends the others (a
    rank that dies would leave its peers waiting in a collective) and its exit code is returned; 0 when every child returned 0.
    Whatever ends the parent early -- an exception, Ctrl-C, SIGTERM -- ends the children too: they hold GPUs."""
    import signal
    procs = start_children(program, argv, child_envs)

    def on_term(signum, _frame):
        raise KeyboardInterrupt("signal %d" % signum)
    previous = None
    try:
        previous = signal.signal(signal.SIGTERM, on_term)
    except ValueError:                                   # not the main thread: no handler, the finally below still runs
        pass
    rc = 0
    live = list(procs)
    kill_at = None
    try:
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
    finally:
        stop_children(procs)
        if previous is not None:
            signal.signal(signal.SIGTERM, previous)
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
