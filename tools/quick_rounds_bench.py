"""bench.py's `rounds` section alone (configs[0] one query, configs[4] 100 weight updates on a resident 10k-clip database)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

if "--profile" in sys.argv:
    import cProfile
    import pstats
    pr = cProfile.Profile()
    pr.enable()
    out = bench.bench_rounds(0, with_cpu=False)
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
else:
    out = bench.bench_rounds(0, with_cpu="--cpu" in sys.argv)
print(json.dumps({k: {q: v[q] for q in ("value", "ms_per_query", "ms_per_round", "ms", "parity_max_abs_err_vs_oracle") if q in v} for k, v in out.items()}, indent=1))
