"""cProfile of bench.py's end-to-end command-line run (where the host's time goes between the file tree and the CSV tree)."""
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

pr = cProfile.Profile()
pr.enable()
r = bench.bench_e2e_cli(0)
pr.disable()
print({k: r[k] for k in ("value", "seconds", "first_run_seconds")})
pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
