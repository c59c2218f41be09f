"""cProfile of bench.py's build_wof_clips end-to-end run (where the host's time goes between the frames and the clip directories)."""
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

bench.bench_e2e_wof(0)
pr = cProfile.Profile()
pr.enable()
r = bench.bench_e2e_wof(0)
pr.disable()
print({k: r[k] for k in ("value", "seconds") if k in r})
pstats.Stats(pr).sort_stats("cumulative").print_stats(35)
