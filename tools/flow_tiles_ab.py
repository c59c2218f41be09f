"""bench.py's TV-L1 object under VQ_FLOW_TILES=square / fitted (the handle reads it at creation): pairs/s and the inner loops' device time."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

for rep in range(2):
    for form in (sys.argv[1:] or ["square", "fitted", "half"]):
        os.environ["VQ_FLOW_TILES"] = form
        r = bench.bench_flow(0, False)
        print(form, json.dumps({k: r[k] for k in ("value", "ms_per_batch", "inner_loops_device_ms_per_batch", "iteration_launches_per_batch") if k in r},
                               default=float), "warped", r.get("warped", {}).get("value") if isinstance(r.get("warped"), dict) else r.get("warped"))
