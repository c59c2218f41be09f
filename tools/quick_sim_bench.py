"""Quick device-side timing of the similarity scan (HIP events through the C ABI)."""
import ctypes as C
import sys
import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import video_query_algorithms_amd as vqa
from video_query_algorithms_amd._lib import call


def bench(n, s, e, d=1024, reps=10, dtype=np.float32):
    db = vqa.FeatureDB.synthetic(n, s, e, d, seed=1, scales=(4.0, 1.0)[:s], dtype=dtype)
    db.set_query_from_row(7, want=False)
    w = [1.0, 1.5][:s]
    db.scan(weights=w)
    tm = C.c_void_p()
    call("vq_timer_create", C.byref(tm))
    times = []
    for _ in range(reps):
        call("vq_timer_start", tm, None)
        db.scan(weights=w)
        call("vq_timer_stop", tm, None)
        ms = C.c_float()
        call("vq_timer_elapsed_ms", tm, C.byref(ms))
        times.append(ms.value)
    nbytes = n * s * e * d * np.dtype(dtype).itemsize + n * 8
    best, med = min(times), sorted(times)[len(times) // 2]
    print("N=%d S=%d E=%d %s: %.3f ms median (%.3f best) -> %.1f GB/s median, %.1f GB/s best (%.2f of 8 TB/s)"
          % (n, s, e, np.dtype(dtype).name, med, best, nbytes / med / 1e6, nbytes / best / 1e6, nbytes / med / 1e6 / 8000),
          flush=True)
    db.close()


if __name__ == "__main__" and len(sys.argv) == 1:
    bench(200_000, 2, 5)
    bench(300_000, 2, 3)
    bench(10_000, 2, 3)
    bench(100_000, 2, 5, dtype=np.float64)
    bench(1_000_000, 2, 5)


def bench_batched(n=1_000_000, s=2, e=5, d=1024, q=16, reps=6):
    """vq_db_scan_batch at cfg 4 (the fused single launch), three timed repeats."""
    import time
    db = vqa.FeatureDB.synthetic(n, s, e, d, seed=17, scales=(4.0, 1.0)[:s])
    rng = np.random.default_rng(0)
    t = rng.standard_normal((q, s, e, d)) / d
    w = 0.5 + rng.random((q, s))
    for form in ("fused", "fused", "fused"):
        db.scan_batch(t, w, want=False)
        db.scores_sync() if hasattr(db, "scores_sync") else db.scan_batch(t[:1], w[:1])     # drain
        tm = C.c_void_p()
        call("vq_timer_create", C.byref(tm))
        times = []
        for _ in range(reps):
            call("vq_timer_start", tm, None)
            db.scan_batch(t, w, want=False)
            call("vq_timer_stop", tm, None)
            ms = C.c_float()
            call("vq_timer_elapsed_ms", tm, C.byref(ms))
            times.append(ms.value)
        med = sorted(times)[len(times) // 2]
        dbb = n * s * e * d * 4
        print("batched Q=%d %s: %.3f ms median (%.3f best) -> %.0f queries/s, DB bytes %.2f TB/s (%.3f of 8 TB/s)"
              % (q, "two-kernel" if form == "1" else "fused", med, min(times), q / med * 1e3, dbb / med / 1e9, dbb / med / 1e9 / 8), flush=True)
    db.close()


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "batched":
    bench_batched()
