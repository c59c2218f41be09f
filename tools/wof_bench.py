import os, sys, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import bench
for mp in ([int(a) for a in sys.argv[1:]] or [None]):
    r = bench.bench_e2e_wof(0, mp)
    print(json.dumps({"max_pairs": mp, "value": r["value"], "seconds": r["seconds"], "first_run_seconds": r["first_run_seconds"]}))
