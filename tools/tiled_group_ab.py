"""One-query scan on the tiled 1M-row database: HIP-event time of a launch (used to pick kTiledGroup, csrc/vq_sim.hip)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import video_query_algorithms_amd as vqa
from video_query_algorithms_amd._lib import call

db = vqa.FeatureDB.synthetic(1_000_000, 2, 5, 1024, seed=17, scales=(4.0, 1.0))
db.set_query_from_row(12345)
tm = C.c_void_p()
call("vq_timer_create", C.byref(tm))
for layout in ("rows", "tiled"):
    db.set_layout(layout)
    ts = []
    for i in range(12):
        call("vq_timer_start", tm, None)
        db.scan(weights=[1.0, 1.5])
        call("vq_timer_stop", tm, None)
        ms = C.c_float()
        call("vq_timer_elapsed_ms", tm, C.byref(ms))
        ts.append(ms.value)
    ts = sorted(ts[2:])
    print("%s: median %.3f ms best %.3f ms -> %.3f of 8 TB/s" % (layout, ts[len(ts) // 2], ts[0], 40.968 / ts[len(ts) // 2] / 8.0), flush=True)
db.close()
