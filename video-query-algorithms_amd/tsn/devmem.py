"""Device buffers and streams for the ingest path, with or without torch.

torch is this package's plumbing for device memory, streams and ``torch.distributed`` -- and ``import torch`` is 0.8 s of a fresh
``python calcSig_wOF.py ...`` that takes 2.3 s for 256 clips.  A single-GPU run of the command line needs four things of a device
runtime (a buffer for a batch's crops, a stream per preparation lane, a wait, a read-back); the library offers them itself
(``vq_dev_malloc`` ...), and this module hands out one or the other:

* ``VQ_NO_TORCH=1`` in the environment and torch not imported yet -- what ``calcSig_wOF.main`` arranges for a one-rank run before the
  library is loaded: the native objects below; torch is never imported, the library runs on the system's HIP runtime;
* otherwise (tests, ``bench.py``, ranks of a multi-GPU run, anything that holds torch tensors): torch, as before.

The decision holds for the whole process: the library must share ONE HIP runtime with whoever else allocates device memory
(``_lib._preload_torch_hip``)."""
from __future__ import annotations

import ctypes as C
import os
import sys
import threading

import numpy as np


def native() -> bool:
    """True when the library's own device plumbing serves this process.  Decided ONCE, when libvqamd.so is loaded (``_lib.TORCHLESS``):
    a later ``import torch`` or a change of the environment cannot flip it mid-process (torch's allocations would then be handed to a
    library that sits on another HIP runtime)."""
    from .. import _lib
    if _lib.TORCHLESS is None:
        _lib.load()
    return bool(_lib.TORCHLESS)


_pool_lock = threading.Lock()
_pool: dict = {}                      # (device, bytes) -> [pointers]: a batch's buffer goes round instead of through hipMalloc / hipFree
_POOL_KEEP = 4                        # (hipFree waits for the whole device; the command line holds 3-4 batches of crops at a time)
_POOL_BYTES = 4 << 30                 # ... and at most this much in all: ragged batch sizes must not pile up a block set per size


class _Block:
    def __init__(self, nbytes: int, device: int):
        from .._lib import call
        self.nbytes, self.device = int(nbytes), int(device)
        with _pool_lock:
            have = _pool.get((self.device, self.nbytes))
            self.ptr = have.pop() if have else None
        if self.ptr is None:
            p = C.c_void_p()
            call("vq_dev_malloc", C.byref(p), self.nbytes, self.device)
            self.ptr = p.value

    def __del__(self):
        # INVARIANT: a block is dropped only after every stream that wrote or read it has been waited for -- vq_jpeg_crops
        # synchronises its lane's stream before it returns the crops, and forward_device(feat_out=...) waits for the network's stream
        # before the batch's crops go out of scope (calcSig_wOF.py's batch loop).  The pool keeps no stream order of its own (torch's
        # caching allocator, which it stands in for, does): a caller that drops a block with work still in flight would let the next
        # batch's decode write into it.  on_device=True consumers must therefore synchronise before releasing.
        try:
            with _pool_lock:
                have = _pool.setdefault((self.device, self.nbytes), [])
                held = sum(k[1] * len(v) for k, v in _pool.items())
                if len(have) < _POOL_KEEP and held + self.nbytes <= _POOL_BYTES:
                    have.append(self.ptr)
                    return
            from .._lib import call
            call("vq_dev_free", C.c_void_p(self.ptr), self.device)
        except Exception:       # interpreter shutdown
            pass


class _HostCopy:
    def __init__(self, a: np.ndarray):
        self._a = a

    def numpy(self) -> np.ndarray:
        return self._a


class DevArray:
    """uint8, C-contiguous, on the device: what the ingest path needs of a tensor (``shape``, ``data_ptr()``, slices along the first
    axis, ``cpu().numpy()``)."""
    is_cuda = True

    def __init__(self, shape, device: int, block: _Block | None = None, offset: int = 0):
        self.shape = tuple(int(v) for v in shape)
        self.device = int(device)
        self._block = block if block is not None else _Block(max(1, int(np.prod(self.shape))), device)
        self._offset = int(offset)

    def data_ptr(self) -> int:
        return self._block.ptr + self._offset

    def __len__(self):
        return self.shape[0]

    def __getitem__(self, key):
        if not isinstance(key, slice):
            raise TypeError("DevArray: slices along the first axis only")
        lo, hi, step = key.indices(self.shape[0])
        if step != 1:
            raise ValueError("DevArray: contiguous slices only")
        row = int(np.prod(self.shape[1:]))
        return DevArray((max(0, hi - lo),) + self.shape[1:], self.device, self._block, self._offset + lo * row)

    def cpu(self):
        from .._lib import call
        out = np.empty(self.shape, dtype=np.uint8)
        if out.size:
            call("vq_dev_read", out.ctypes.data_as(C.c_void_p), C.c_void_p(self.data_ptr()), out.size, self.device)
        return _HostCopy(out)


class Stream:
    """A non-blocking HIP stream of the library's (``cuda_stream``: its handle, as torch names it)."""

    def __init__(self, device: int, priority: int = 0):
        from .._lib import call
        p = C.c_void_p()
        if priority:
            call("vq_stream_create_priority", C.byref(p), int(device), int(priority))
        else:
            call("vq_stream_create", C.byref(p), int(device))
        self.cuda_stream, self.device = p.value, int(device)

    def synchronize(self):
        from .._lib import call
        call("vq_stream_synchronize", C.c_void_p(self.cuda_stream), self.device)

    def __del__(self):
        try:
            from .._lib import call
            call("vq_stream_destroy", C.c_void_p(self.cuda_stream), self.device)
        except Exception:
            pass


def empty_u8(shape, device: int):
    """An uninitialised uint8 device array of ``shape``: a torch tensor, or a DevArray in a process that runs without torch."""
    if native():
        return DevArray(shape, device)
    import torch
    return torch.empty(tuple(shape), dtype=torch.uint8, device=torch.device("cuda", int(device)))


def new_stream(device: int, priority: int = 0):
    """A non-blocking stream; ``priority`` -1 / 0 / +1: the lowest queue priority of the device / the default / the highest (the
    library's own stream in every case but the default one under torch: torch offers no priority below its default)."""
    if native() or priority:
        return Stream(device, priority)
    import torch
    return torch.cuda.Stream(device=torch.device("cuda", int(device)))


def current_stream_handle(device: int) -> int:
    """The stream resize / crop work of the host-frames path is queued on: torch's current stream, or the default stream."""
    if native():
        return 0
    import torch
    return torch.cuda.current_stream(torch.device("cuda", int(device))).cuda_stream


def synchronize_current(device: int):
    if native():
        from .._lib import call
        call("vq_stream_synchronize", None, int(device))
        return
    import torch
    torch.cuda.current_stream(torch.device("cuda", int(device))).synchronize()
