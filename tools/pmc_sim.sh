#!/bin/bash
# HBM traffic of the similarity scan from the L2 memory-side counters (separate --pmc passes, kernel-trace only).
set -e
mkdir -p gpurun_out
cat > /tmp/sim_once.py <<'PY'
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import video_query_algorithms_amd as vqa
db = vqa.FeatureDB.synthetic(1_000_000, 2, 5, 1024, seed=17, scales=(4.0, 1.0))
db.set_query_from_row(12345, want=False)
for _ in range(3):
    db.scan(weights=[1.0, 1.5])
db.scores()
PY
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc_sim_fetch --output-format csv -- python3 /tmp/sim_once.py > gpurun_out/pmc_sim_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmc_sim_write --output-format csv -- python3 /tmp/sim_once.py > gpurun_out/pmc_sim_write.log 2>&1
ls gpurun_out/pmc_sim_fetch/*/ gpurun_out/pmc_sim_write/*/
