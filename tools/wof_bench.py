import os, sys, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import bench
print(json.dumps(bench.bench_e2e_wof(0)))
