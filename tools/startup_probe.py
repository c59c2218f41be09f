"""Where a fresh process's first second goes: loading the library and the first HIP call, with and without torch in the process.
    python tools/startup_probe.py [torch]"""
import os
import sys
import time

t0 = time.perf_counter()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "torch":
    import torch      # noqa: F401
    t_torch = time.perf_counter()
    print("import torch            %.3f s" % (t_torch - t0))
else:
    os.environ["VQ_NO_TORCH"] = "1"
t1 = time.perf_counter()
import numpy as np    # noqa: E402,F401
from video_query_algorithms_amd import _lib   # noqa: E402
t2 = time.perf_counter()
print("numpy + package         %.3f s" % (t2 - t1))
lib = _lib.load()
t3 = time.perf_counter()
print("library loaded          %.3f s" % (t3 - t2))
import ctypes as C    # noqa: E402
p = C.c_void_p()
_lib.call("vq_dev_malloc", C.byref(p), 1 << 20, 0)
t4 = time.perf_counter()
print("first HIP call (malloc) %.3f s" % (t4 - t3))
_lib.call("vq_dev_malloc", C.byref(p), 1 << 30, 0)
_lib.call("vq_stream_synchronize", None, 0)
t5 = time.perf_counter()
print("1 GB malloc + sync      %.3f s" % (t5 - t4))
print("total                   %.3f s (torch imported: %s)" % (t5 - t0, "torch" in sys.modules))
