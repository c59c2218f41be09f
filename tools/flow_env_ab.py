"""bench.py's TV-L1 object under values of ONE environment variable the flow handle reads at creation:  tools/flow_env_ab.py VQ_FLOW_CHUNK 4 2 1"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

name, values = sys.argv[1], sys.argv[2:]
for rep in range(2):
    for v in values:
        os.environ[name] = v
        r = bench.bench_flow(0, False)
        print(name + "=" + v, json.dumps({k: r[k] for k in ("value", "ms_per_batch", "inner_loops_device_ms_per_batch", "iteration_launches_per_batch") if k in r}, default=float),
              "warped", r.get("warped"))
