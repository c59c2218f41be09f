"""bench.py's end-to-end command-line object with VQ_CLI_TRACE=1: per stream, where the batch loop waited (stderr)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["VQ_CLI_TRACE"] = "1"
import bench

r = bench.bench_e2e_cli(0)
print(json.dumps({k: r[k] for k in ("value", "seconds", "first_run_seconds", "seconds_32_clips", "steady_state")}))
